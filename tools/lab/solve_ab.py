"""Whole solves on ONE box with a tuning knob toggled (same resident hierarchy): python tools/lab/solve_ab.py <n> <key> <v0,v1,...> [var] [reps]
-> ms per solve for every value, interleaved over `reps` rounds (box-to-box spread is larger than most effects measured this way)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]); key = sys.argv[2].encode(); vals = [int(v) for v in sys.argv[3].split(",")]
var = len(sys.argv) > 4 and sys.argv[4] == "var"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
if var:
    ia, ja, a, f = fa.poisson7pt_var(n, (ia, ja, a, f, ue))
itp = fa.param_solver_init(); itp.tol = 1e-8; itp.maxit = 500; itp.print_level = 0
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
H.set_rhs(f)
res = {v: [] for v in vals}
for r in range(reps + 1):
    for v in vals:
        L.fasp_hip_tune(key, v)
        st, hist, stats = H.solve_resident(itp)
        L.fasp_hip_device_synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            st, hist, stats = H.solve_resident(itp)
        L.fasp_hip_device_synchronize()
        if r:
            res[v].append((time.perf_counter() - t0) / 5 * 1e3)
for v in vals:
    print(f"{key.decode()} = {v}: {np.mean(res[v]):.2f} ms per solve (rounds: {' '.join(f'{x:.2f}' for x in res[v])}), {st} iterations, relres {stats.relres:.6e}")
H.close()
