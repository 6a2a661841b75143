"""Is the long-row kernel gather-bound?  Timing experiments on levels 3..9 of P7(256) (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(256)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
for lev in range(3, H.num_levels):
    r, c, _, _, v = H.matrix(lev, 0)
    line = f"L{lev} rows {r:7d} nnz {len(v):9d}:"
    for dbg in (0, 98, 99):
        L.fasp_hip_tune(b"dbg", dbg)
        ms = H.time_kernel(0, lev, 30)
        line += f" dbg{dbg} {ms*1e3:6.1f}us ({12*len(v)/ms/1e6:5.0f} GB/s) |"
    print(line, flush=True)
L.fasp_hip_tune(b"dbg", 0)
H.close()
