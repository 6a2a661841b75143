"""What does a gather cost in the coded square kernel?  Stencil subsets of P7(n) (same rows, fewer entries per row)
through the resident upload path: time per launch of y = A x.  usage: python tools/expt_stencil_subsets.py [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ia, ja, a, f, ue = fa.poisson7pt(n)
rows = np.repeat(np.arange(len(f), dtype=np.int64), np.diff(ia))
off = ja.astype(np.int64) - rows
s1, s2 = n, n * n
subsets = {"7pt (0,+-1,+-n,+-n2)": None, "5: 0,+-1,+-n": (0, 1, -1, s1, -s1), "5: 0,+-n,+-n2 (aligned)": (0, s1, -s1, s2, -s2),
           "5: 0,+-1,+-n2": (0, 1, -1, s2, -s2), "3: 0,+-1": (0, 1, -1), "3: 0,+-n": (0, s1, -s1), "3: 0,+-n2": (0, s2, -s2), "1: diagonal": (0,)}
L = fa.lib()
for name, keep in subsets.items():
    if keep is None:
        ia2, ja2, a2 = ia, ja, a
    else:
        m = np.isin(off, np.array(keep))
        ja2 = ja[m].copy(); a2 = a[m].copy()
        cnt = np.bincount(rows[m], minlength=len(f))
        ia2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    A, keepalive = T.as_csr(ia2, ja2, a2)
    kind = C.c_int(-1)
    line = f"{name:28s} nnz/row {len(a2)/len(f):4.2f}:"
    for op, nm in ((0, "mxv"), (5, "mxv+dot"), (2, "jacobi")):
        ms = L.fasp_hip_time_matrix(C.byref(A), op, 30, C.byref(kind))
        line += f" {nm} {ms*1e3:6.1f} us |"
    print(line, f"kernel family {kind.value}", flush=True)
