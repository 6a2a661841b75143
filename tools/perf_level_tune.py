"""Kernel times of one level under values of one fasp_hip_tune key (dev tool):
python tools/perf_level_tune.py n level key v1 v2 ...   (ops: mxv, jacobi, mxv+dot)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]); lev = int(sys.argv[2]); key = sys.argv[3].encode(); vals = [int(v) for v in sys.argv[4:]]
L = fa.lib()
if os.environ.get("VAR"):
    ia, ja, a, f = fa.poisson7pt_var(n)
else:
    ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
for v in vals:
    L.fasp_hip_tune(key, v)
    t = [min(H.time_kernel(k, lev, 20) for _ in range(2)) * 1e3 for k in (0, 2, 5)]
    tr = [min(H.time_kernel(k, lev, 20) for _ in range(2)) * 1e3 for k in (6, 7)] if lev < H.num_levels - 1 else [0, 0]
    print(f"level {lev} kind {H.kernel_info(lev, 0)[0]} {key.decode()} = {v:5d}: mxv {t[0]:7.1f} us  jacobi {t[1]:7.1f} us  mxv+dot {t[2]:7.1f} us | R mxv {tr[0]:7.1f} us (kind {H.kernel_info(lev, 2)[0] if lev < H.num_levels - 1 else -1})  P aAxpy {tr[1]:7.1f} us (kind {H.kernel_info(lev, 1)[0] if lev < H.num_levels - 1 else -1})", flush=True)
H.close()
