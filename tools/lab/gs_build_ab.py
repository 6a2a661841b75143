"""GS-default solve of P7(n) for several values of a tune key that is read when the sweep schedules are BUILT (a new hierarchy per value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
key = sys.argv[2].encode()
vals = [int(v) for v in sys.argv[3].split(",")]
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
for v in vals:
    L.fasp_hip_tune(key, v)
    amgp = fa.param_amg_init()
    mode = os.environ.get('MODE', 'cf')
    if mode == 'nat': amgp.smooth_order = 0
    if mode == 'sor': amgp.smoother = T.SMOOTHER_SOR; amgp.relaxation = 1.1; amgp.smooth_order = 0
    H = fa.AMG(ia, ja, a, amgp); H.set_rhs(f)
    ts = []
    for _ in range(4):
        st, hist, stats = H.solve_resident(itp)
        ts.append(stats.solve_seconds * 1e3)
    print(f"{key.decode()} {v}: {st} iterations, relres {stats.relres:.10e}, solve best {min(ts[1:]):.1f} mean {np.mean(ts[1:]):.1f} ms", flush=True)
    H.close()
