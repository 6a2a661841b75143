"""Which part of k_csr_estream costs what (lab build with -DFASP_LAB_DEBUG: tools/build_variant.sh dbg -DFASP_LAB_DEBUG, run with
FASP_HIP_LIB=lab_build/libfasp_hip_dbg.so): y = A x on the long-row levels, cold, with parts switched off --
es_dbg 1 no gathers (x := 1), 2 no stream loads (the slab is processed as it stands), 3 both, 4 no row sums, 7 nothing but the loop."""
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
print("us per launch cold, y = A x: full | no gathers | no stream | neither | no row sums | loop only")
for l in levels:
    row = []
    for d in (0, 1, 2, 3, 4, 7):
        L.fasp_hip_tune(b"es_dbg", d)
        row.append(H.time_kernel(0, l, 6) * 1e3)
    print(f"level {l}: " + " | ".join(f"{x:6.1f}" for x in row), flush=True)
L.fasp_hip_tune(b"es_dbg", 0); L.fasp_hip_tune(b"time_cold", 0)
H.close()
