"""Coarse safe CG: fused one-launch iteration vs SpMV + step kernel, grid sweep, on P7(n) (dev tool)."""
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
def run(tag):
    for rep in range(3):
        st, hist, stats = H.solve_resident(itp)
    print(f"{tag:28s}: iters {st} relres {stats.relres:.10e} solve {stats.solve_seconds*1e3:.2f} ms coarse its {stats.coarse_iters}", flush=True)
L.fasp_hip_tune(b"spcg_fused", 0); run("two kernels, batch 8")
L.fasp_hip_tune(b"spcg_fused", 1)
grids = [int(x) for x in sys.argv[2:]] or [0, 255, 311, 415, 511, 622, 767, 1023]
for g in grids:
    L.fasp_hip_tune(b"spcg_grid", g)
    for b in (8, 16):
        L.fasp_hip_tune(b"spcg_batch", b)
        run(f"fused grid {g} batch {b}")
H.close()
