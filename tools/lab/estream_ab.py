"""Long-row levels of the P7(n) hierarchy, COLD (one launch behind a 512 MB read, as a V-cycle meets them): the row kernel
(k_csr_rows) against the entry-parallel stream (k_csr_estream) -- SpMV / y -= A x / Jacobi,
us per launch, and the plain 16-byte read of the level's values beside them.   python tools/lab/estream_ab.py [n] [var]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
var = len(sys.argv) > 2 and sys.argv[2] == "var"
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
if var:
    ia, ja, a, f = fa.poisson7pt_var(n, (ia, ja, a, f, ue))
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
H = fa.AMG(ia, ja, a, amgp)
L.fasp_hip_tune(b"time_cold", 1)
levels = [l for l in range(H.num_levels - 1) if H.kernel_info(l, 0)[0] == 0]
print(f"P7({n}){' variable' if var else ''}: long-row levels {levels}; us per launch cold: SpMV / y -= A x / Jacobi   (plain read of the values: last column)")
for l in levels:
    kind, mb = H.kernel_info(l, 0)
    row = []
    for es in (0, 2):
        L.fasp_hip_tune(b"estream", es)
        row.append([H.time_kernel(k, l, 8) * 1e3 for k in (0, 1, 2)])
    rd = H.time_kernel(8, l, 8) * 1e3
    r, c, *_ = H.matrix(l, 0)
    print(f"level {l}: {r:8d} rows {mb/1e6:7.1f} MB | rows " + " / ".join(f"{x:6.1f}" for x in row[0]) + " | estream " + " / ".join(f"{x:6.1f}" for x in row[1])
          + f" | read {rd:6.1f}", flush=True)
L.fasp_hip_tune(b"estream", 1); L.fasp_hip_tune(b"time_cold", 0)
H.close()
