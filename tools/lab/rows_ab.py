"""A/B of a tune key on the long-row levels of P7(n): SpMV / Jacobi per level, back to back and cold (time_cold), and the solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
key = (sys.argv[2] if len(sys.argv) > 2 else "ja16").encode()
vals = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
itp = fa.param_solver_init(); itp.tol = 1e-8
H.set_rhs(f)
for rep in range(2):
    for v in vals:
        L.fasp_hip_tune(key, v)
        ts, sp = [], []
        for _ in range(6):
            st, hist, stats = H.solve_resident(itp)
            ts.append(stats.solve_seconds * 1e3); sp.append(stats.spmv_ms * 1e3)
        out = []
        for cold in (0, 1):
            L.fasp_hip_tune(b"time_cold", cold)
            out.append(" ".join(f"L{l}:{H.time_kernel(0, l, 10) * 1e3:.0f}/{H.time_kernel(2, l, 10) * 1e3:.0f}" for l in range(2, H.num_levels - 1)))
        L.fasp_hip_tune(b"time_cold", 0)
        print(f"{key.decode()} {v}: iters {st} relres {stats.relres:.10e} solve best {min(ts):.2f} mean {np.mean(ts[1:]):.2f} ms, t = A p inside the solve {np.mean(sp[1:]):.1f} us\n    SpMV/Jacobi us back to back: {out[0]}\n    cold: {out[1]}", flush=True)
H.close()
