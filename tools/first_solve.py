"""First and warm solve times of P7(n) with the sequential smoothers (sweep schedules built behind the setup): python tools/first_solve.py n"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1])
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
for name, mod in (("SOR 1.1 natural", lambda p: (setattr(p, "smoother", T.SMOOTHER_SOR), setattr(p, "relaxation", 1.1), setattr(p, "smooth_order", 0))),
                  ("GS natural", lambda p: setattr(p, "smooth_order", 0)), ("SGS", lambda p: setattr(p, "smoother", T.SMOOTHER_SGS)), ("GS-CF", lambda p: None)):
    p = fa.param_amg_init(); mod(p)
    t = time.time(); H = fa.AMG(ia, ja, a, p); ts = time.time() - t
    H.set_rhs(f)
    out = []
    for r in range(3):
        st, hist, stats = H.solve_resident(itp); out.append(f"{stats.solve_seconds*1e3:.1f}")
    print(f"P7({n}) {name}: setup {ts:.2f} s, {st} iterations, relres {stats.relres:.6e}, solves {' / '.join(out)} ms", flush=True)
    H.close()
