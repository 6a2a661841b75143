"""Every level of the P7(n) hierarchy: SpMV / residual-like axpy / Jacobi / R / P, back to back, behind 512 MB read (cold) and behind 512 MB written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
tot = {}
for cold, name in ((0, "back to back"), (1, "behind 512 MB read"), (2, "behind 512 MB written")):
    L.fasp_hip_tune(b"time_cold", cold)
    print(f"--- {name}: us per launch, SpMV / y -= A x / Jacobi / R / P / plain read of the values")
    s = 0.0
    for l in range(H.num_levels - 1):
        t = [H.time_kernel(k, l, 10) * 1e3 for k in (0, 1, 2, 6, 7, 8)]
        s += t[1] + t[2] + t[3] + t[4]
        kind, mb = H.kernel_info(l, 0)
        print(f"level {l} kind {kind:2d} {mb / 1e6:7.1f} MB: " + " / ".join(f"{x:6.1f}" for x in t), flush=True)
    print(f"    one residual + one Jacobi + R + P over the levels: {s:.0f} us")
L.fasp_hip_tune(b"time_cold", 0)
H.close()
