"""Coarse safe CG variants on the P7(n) hierarchy: whole-solve time with the persistent register-resident kernel
(spcg_persist 1), the one-launch-per-iteration kernel (0) -- development tool.  usage: perf_coarse.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
itp = fa.param_solver_init(); itp.tol = 1e-8; itp.print_level = 0
H = fa.AMG(ia, ja, a, amgp)
H.set_rhs(f)
L = fa.lib()
for persist in (1, 0, 1, 0):
    L.fasp_hip_tune(b"spcg_persist", persist)
    H.solve_resident(itp)
    L.fasp_hip_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st, hist, stats = H.solve_resident(itp)
    L.fasp_hip_device_synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"spcg_persist {persist}: {dt*1e3:.2f} ms/solve, {st} iterations, relres {stats.relres:.10e}, coarse iterations {stats.coarse_iters}", flush=True)
H.close()
