"""Reference DEFAULT parameters (GS smoother with C/F ordering) and SOR on P7(n): solve time with the persistent sweep
kernel (seq_persist 1: one launch per sweep) and with one launch per dependency level (0).  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
for name, mod in (("GS-CF (defaults)", lambda p: None), ("SOR w=1.1, natural order", lambda p: (setattr(p, "smoother", T.SMOOTHER_SOR), setattr(p, "relaxation", 1.1), setattr(p, "smooth_order", 0)))):
    amgp = fa.param_amg_init(); mod(amgp)
    t = time.time(); H = fa.AMG(ia, ja, a, amgp); print(f"P7({n}) {name}: setup {time.time()-t:.1f}s levels {H.num_levels}", flush=True)
    H.set_rhs(f)
    for persist in (1, 0, 1, 0):
        fa.lib().fasp_hip_tune(b"seq_persist", persist)
        st, hist, stats = H.solve_resident(itp)
        print(f"  seq_persist {persist}: iters {st} relres {stats.relres:.6e} solve {stats.solve_seconds*1e3:.1f} ms coarse its {stats.coarse_iters}", flush=True)
    H.close()
fa.lib().fasp_hip_tune(b"seq_persist", 1)
