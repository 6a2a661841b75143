"""In-process A/B of the streaming (nt) hints of the coded pair kernels: off / automatic (vectors > 96 MB) / everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
H.set_rhs(f)
itp = fa.param_solver_init(); itp.tol = 1e-8
L = fa.lib()
for rnd in range(3):
    for mode, name in ((0, "off "), (-1, "auto"), (1, "all ")):
        L.fasp_hip_tune(b"rp_stream", mode)
        ts = []
        for rep in range(4):
            st, hist, stats = H.solve_resident(itp)
            ts.append(stats.solve_seconds * 1e3)
        print(f"rp_stream {name}: solve {min(ts):.2f} ms (min of 4), level-0 t = A p {stats.spmv_ms*1e3:.1f} us", flush=True)
L.fasp_hip_tune(b"rp_stream", -1)
H.close()
