import sys; sys.path.insert(0,'/root/repo')
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n=int(sys.argv[1])
ia,ja,a,f,ue=fa.poisson7pt(n)
p=fa.param_amg_init(); p.smoother=T.SMOOTHER_JACOBI; p.relaxation=0.6667
H=fa.AMG(ia,ja,a,p)
L=fa.lib()
for l in range(3):
    print("level",l,"kinds A,P,R:",[H.kernel_info(l,w) for w in (0,1,2)],flush=True)
for g in (2,1,2,1):
    L.fasp_hip_tune(b"gen2",g)
    print("gen2",g," ".join(f"L{l}: R {H.time_kernel(6,l,20)*1e3:.1f} P {H.time_kernel(7,l,20)*1e3:.1f}" for l in (0,1)),flush=True)
