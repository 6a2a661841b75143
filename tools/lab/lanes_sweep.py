"""Lanes per row of k_csr_rows on the long-row levels, with the brick renumbering on: SpMV / Jacobi us per level for lanes 8..64 (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
H.set_rhs(f)
for lanes in (-1, 8, 16, 32, 64):
    L.fasp_hip_tune(b"lanes", lanes)
    row = " ".join(f"L{l}:{min(H.time_kernel(0, l, 20) for _ in range(2)) * 1e3:.1f}/{min(H.time_kernel(1, l, 20) for _ in range(2)) * 1e3:.1f}" for l in range(3, H.num_levels))
    print(f"lanes {lanes:3d}: SpMV / residual us  {row}", flush=True)
L.fasp_hip_tune(b"lanes", -1)
