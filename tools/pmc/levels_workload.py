"""Workload of tools/pmc/run_levels.sh: the A-operator kernels of the P7(n) hierarchy, level by level, a k_dot launch
between two levels as a separator in the dispatch order (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa  # noqa: E402
from faspsolver_amd import _types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ops = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
for l in range(H.num_levels):
    for k in ops:
        if k in (6, 7) and l == H.num_levels - 1:
            continue
        ms = H.time_kernel(k, l, reps)
        print(f"level {l} op {k}: {ms*1e3:.1f} us", flush=True)
        H.time_kernel(3, H.num_levels - 1, 1)   # separator
H.close()
