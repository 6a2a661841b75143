"""A/B of a tune key on the GS-default solve of P7(n) (same handle, key read at launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import faspsolver_amd as fa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
key = (sys.argv[2] if len(sys.argv) > 2 else "seq_chain_touch").encode()
vals = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8]
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
H = fa.AMG(ia, ja, a, fa.param_amg_init()); H.set_rhs(f)
H.solve_resident(itp)
for rep in range(3):
    for v in vals:
        L.fasp_hip_tune(key, v)
        ts = []
        for _ in range(3):
            st, hist, stats = H.solve_resident(itp)
            ts.append(stats.solve_seconds * 1e3)
        print(f"{key.decode()} {v}: {st} iterations, relres {stats.relres:.10e}, solve best {min(ts):.1f} mean {np.mean(ts):.1f} ms", flush=True)
H.close()
