"""In-kernel section timing of k_csr_estream (lab build: tools/build_variant.sh est -DFASP_LAB_DEBUG -DES_TIMING; FASP_HIP_LIB=lab_build/libfasp_hip_est.so)."""
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
for l in (3, 7):
    print(f"--- level {l}", flush=True)
    print(f"{H.time_kernel(0, l, 1) * 1e3:.1f} us", flush=True)
L.fasp_hip_tune(b"time_cold", 0)
H.close()
