"""Coarse safe-CG batch size sweep on P7(n) (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
itp = fa.param_solver_init(); itp.tol = 1e-8
H.set_rhs(f)
for b in (1, 8, 16, 32):
    L.fasp_hip_tune(b"spcg_batch", b)
    for rep in range(3):
        st, hist, stats = H.solve_resident(itp)
    print(f"batch {b:3d}: iters {st} relres {stats.relres:.10e} solve {stats.solve_seconds*1e3:.2f} ms coarse its {stats.coarse_iters}", flush=True)
H.close()
