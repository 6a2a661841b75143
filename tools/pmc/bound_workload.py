"""Workload of tools/pmc/run_bound.sh: every level's SpMV of the P7(n) hierarchy COLD (one launch behind a 512 MB read of
other data, as a V-cycle meets it), next to a plain 16-byte-per-lane read of the same level's values; a k_dot launch
separates two (level, kind) segments in the dispatch order (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa  # noqa: E402
from faspsolver_amd import _types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kinds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,8").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
var = len(sys.argv) > 4 and sys.argv[4] == "var"
L = fa.lib()
if var:
    L.fasp_hip_tune(b"compress", 0)
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
L.fasp_hip_tune(b"time_cold", 1)
seg = 0
for l in range(H.num_levels - 1):
    for k in kinds:
        ms = H.time_kernel(k, l, reps)
        kind, mb = H.kernel_info(l, 0)
        print(f"segment {seg} level {l} op {k} kernel-kind {kind} bytes {mb:.0f}: {ms*1e3:.1f} us", flush=True)
        L.fasp_hip_tune(b"time_cold", 0)
        H.time_kernel(3, H.num_levels - 1, 1)   # separator (three k_dot launches)
        L.fasp_hip_tune(b"time_cold", 1)
        seg += 1
H.close()
