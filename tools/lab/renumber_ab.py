"""A/B of the brick renumbering (fasp_hip_tune("renumber")): solve time of the headline and of its variable-coefficient twin, per-level kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
for var in (0,):
    ia, ja, a, f, ue = fa.poisson7pt(n)
    if var:
        ia, ja, a, f = fa.poisson7pt_var(n, (ia, ja, a, f, ue))
    for ren in (1, 2, 1, 2):
        L.fasp_hip_tune(b"renumber", ren)
        amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
        import time
        t0 = time.time()
        H = fa.AMG(ia, ja, a, amgp)
        ts = time.time() - t0
        itp = fa.param_solver_init(); itp.tol = 1e-8
        H.set_rhs(f)
        best = 1e9
        for rep in range(5):
            st, hist, stats = H.solve_resident(itp)
            best = min(best, stats.solve_seconds)
        lv = " ".join(f"L{l}:k{H.kernel_info(l, 0)[0]}:{H.time_kernel(0, l, 10) * 1e3:.0f}/{H.time_kernel(2, l, 10) * 1e3:.0f}/R{H.time_kernel(6, l, 10) * 1e3:.0f}/P{H.time_kernel(7, l, 10) * 1e3:.0f}" for l in range(min(4, H.num_levels - 1)))
        print(f"{'variable' if var else 'P7      '} renumber {ren}: setup {ts:.1f} s, iters {st} relres {stats.relres:.10e} solve {best*1e3:.2f} ms | SpMV/Jacobi us per level: {lv}", flush=True)
        H.close()
L.fasp_hip_tune(b"renumber", 1)
