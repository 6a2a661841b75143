"""R and P of the coded levels 0 and 1 of P7(n): us per launch back to back and cold (fasp_hip_time_kernel kinds 6, 7)."""
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
ts = []
for _ in range(6):
    st, hist, stats = H.solve_resident(itp)
    ts.append(stats.solve_seconds * 1e3)
print(f"iters {st} relres {stats.relres:.10e} solve best {min(ts):.2f} mean {np.mean(ts[1:]):.2f} ms")
for rp5 in (45, 200):
  L.fasp_hip_tune(b"rp5_max", rp5)
  print(f"rp5_max {rp5}")
  for cold in (0, 1):
    L.fasp_hip_tune(b"time_cold", cold)
    print(("cold        " if cold else "back to back") + ": " + "  ".join(f"L{l}: R {H.time_kernel(6, l, 10) * 1e3:.1f} (kind {H.kernel_info(l, 2)[0]}) P {H.time_kernel(7, l, 10) * 1e3:.1f} (kind {H.kernel_info(l, 1)[0]})" for l in range(0, 3)), flush=True)
L.fasp_hip_tune(b"time_cold", 0); L.fasp_hip_tune(b"rp5_max", 45)
H.close()
