"""End-to-end time of the drop-in call fasp_solver_dcsr_krylov_amg (setup + upload + solve + download) (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ia, ja, a, f, ue = fa.poisson7pt(n)
itp, amgp = fa.param_solver_init(), fa.param_amg_init()
itp.tol = 1e-8; itp.print_level = 2; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667; amgp.print_level = 1
for rep in range(2):
    x = np.zeros(len(f))
    t0 = time.time()
    st = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, itp, amgp)
    print(f"drop-in call P7({n}): status {st}, {time.time()-t0:.2f} s end to end, max|x-u| {np.abs(x-ue).max():.3e}", flush=True)
