"""Reference DEFAULT parameters (GS smoother with C/F ordering) on P7(n): solve time (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
amgp = fa.param_amg_init()
t = time.time(); H = fa.AMG(ia, ja, a, amgp); print(f"P7({n}) setup {time.time()-t:.1f}s levels {H.num_levels}", flush=True)
H.set_rhs(f)
for rep in range(2):
    st, hist, stats = H.solve_resident(itp)
    print(f"defaults (GS-CF): iters {st} relres {stats.relres:.6e} solve {stats.solve_seconds*1e3:.1f} ms coarse its {stats.coarse_iters}", flush=True)
H.close()
