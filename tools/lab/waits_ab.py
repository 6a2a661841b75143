"""A/B of the host round trips of one PCG iteration on the headline solve (tune keys pcg_dev_beta, spcg_spec, ev_every)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
cases = [("all as before", 0, 0, 1, 0), ("beta on the device", 1, 0, 1, 0), ("+ one wait per coarse solve", 1, 1, 1, 0), ("+ an event pair every 4th iteration", 1, 1, 4, 0), ("+ (t,p), (z,r) summed by their consumers", 1, 1, 4, 1)]
for rep in range(3):
    for name, b, sp, ev, fold in cases:
        L.fasp_hip_tune(b"pcg_dev_beta", b); L.fasp_hip_tune(b"spcg_spec", sp); L.fasp_hip_tune(b"ev_every", ev); L.fasp_hip_tune(b"pcg_fold", fold)
        ts = []
        for _ in range(8):
            st, hist, stats = H.solve_resident(itp)
            ts.append(stats.solve_seconds * 1e3)
        print(f"{name:40s}: iters {st} relres {stats.relres:.10e} solve best {min(ts):.2f} mean {np.mean(ts[2:]):.2f} ms", flush=True)
L.fasp_hip_tune(b"pcg_dev_beta", 1); L.fasp_hip_tune(b"spcg_spec", 1); L.fasp_hip_tune(b"ev_every", 4); L.fasp_hip_tune(b"pcg_fold", 1)
H.close()
