"""Solve time of P7(n) under values of one fasp_hip_tune key (dev tool): python tools/perf_tune.py n key v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]); key = sys.argv[2].encode(); vals = [int(v) for v in sys.argv[3:]]
L = fa.lib()
if os.environ.get("VAR"):
    ia, ja, a, f = fa.poisson7pt_var(n)   # variable coefficients: every level on the plain-CSR kernels
else:
    ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
itp = fa.param_solver_init(); itp.tol = 1e-8
H.set_rhs(f)
for v in vals:
    L.fasp_hip_tune(key, v)
    best = 1e9
    for rep in range(4):
        st, hist, stats = H.solve_resident(itp)
        best = min(best, stats.solve_seconds)
    print(f"{key.decode()} = {v:5d}: iters {st} relres {stats.relres:.10e} solve {best*1e3:.2f} ms coarse its {stats.coarse_iters}", flush=True)
H.close()
