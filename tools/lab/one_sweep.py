import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/faspsolver_amd') else os.getcwd())
import faspsolver_amd as fa
n = int(sys.argv[1]); lev = int(sys.argv[2]); kind = int(sys.argv[3])
ia, ja, a, f, ue = fa.poisson7pt(n)
H = fa.AMG(ia, ja, a, fa.param_amg_init())
H.set_rhs(f)
if len(sys.argv) > 4: fa.lib().fasp_hip_tune(b"time_cold", int(sys.argv[4]))
print("time", H.time_kernel(kind, lev, 1) * 1e3, "us", flush=True)
