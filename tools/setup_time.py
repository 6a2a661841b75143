"""Host AMG setup of P7(n) with the phase timers on (FASP_HIP_SETUP_TIMING=1): python tools/setup_time.py [n] [host_only]"""
import os, sys, time
os.environ.setdefault("FASP_HIP_SETUP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ia, ja, a, f, ue = fa.poisson7pt(n)
p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667
t = time.time()
H = fa.AMG(ia, ja, a, p, host_only=len(sys.argv) > 2)
print(f"setup total {time.time() - t:.2f} s, levels {H.num_levels}", flush=True)
