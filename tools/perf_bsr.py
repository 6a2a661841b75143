"""BSR SpMV bandwidth on the synthetic config-3 matrix P7(n) (x) B3 and on SPE01 (dev tool)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
import _libs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = fa.lib()
for name, (ia, ja, val, nb) in (("SPE01", _libs.read_bsr(_libs.DATA + "/bsrmat_SPE01.dat")),
                                (f"P7({n})xB3", None)):
    if ia is None:
        ia, ja, a, f, ue = fa.poisson7pt(n)
        nb = 3
        val = (a[:, None, None] * _libs.B3[None, :, :]).reshape(-1)
    A, keep = T.as_bsr(ia, ja, val, nb)
    ms = L.fasp_hip_time_bsr_mxv(C.byref(A), 20)
    B = A.NNZ * (8 * nb * nb + 4) + 4 * (A.ROW + 1) + 16 * A.ROW * nb
    print(f"{name}: ROW {A.ROW} NNZ {A.NNZ} nb {nb}: {ms*1e3:.1f} us/launch, {B/ms/1e6:.0f} GB/s algorithmic "
          f"({B/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
