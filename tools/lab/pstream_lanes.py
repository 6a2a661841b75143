"""k_csr_pstream: lanes per sub-row forced (fasp_hip_tune("ps_lanes")), cold y = A x on the long-row levels.  python tools/lab/pstream_lanes.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
L.fasp_hip_tune(b"time_cold", 1)
levels = [l for l in range(H.num_levels - 1) if H.kernel_info(l, 0)[0] == 0]
print("us per launch cold, y = A x (pstream + combine), lanes per sub-row = auto / 2 / 4 / 8 / 16 / 64 (register form)")
for l in levels:
    row = []
    for ln in (0, 2, 4, 8, 16, 64):
        L.fasp_hip_tune(b"ps_lanes", ln)
        row.append(H.time_kernel(0, l, 6) * 1e3)
    print(f"level {l}: " + " | ".join(f"{x:6.1f}" for x in row), flush=True)
L.fasp_hip_tune(b"ps_lanes", 0); L.fasp_hip_tune(b"time_cold", 0)
H.close()
