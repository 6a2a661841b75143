"""What does a brick-like renumbering of a coarse level buy?  Level `lev` of the P7(n) hierarchy (VAR=1: the
variable-coefficient twin) in natural order and in the cluster order of reorder.cpp, through the resident upload
path (k_csr_wstream2 / k_csr_xtile as upload_csr selects them): us per launch of y = A x and of a Jacobi sweep."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2").split(",")]
var = os.environ.get("VAR") == "1"
ia, ja, a, f, ue = fa.poisson7pt(n)
if var:
    _i, _j, a, f = fa.poisson7pt_var(n, (ia, ja, a, f, ue))
p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667
H = fa.AMG(ia, ja, a, p, host_only=True)
L = fa.lib()
def t(M):
    M = M.tocsr()
    A, keep = T.as_csr(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data)
    k = C.c_int(0)
    out = []
    for xt in (1, 0):
        L.fasp_hip_tune(b"xtile", xt)
        out.append(L.fasp_hip_time_matrix(C.byref(A), 0, 20, C.byref(k)) * 1e3)
    L.fasp_hip_tune(b"xtile", 1)
    rows = []
    for lanes in (4, 8, 16):
        L.fasp_hip_tune(b"kind", 0); L.fasp_hip_tune(b"lanes", lanes)
        rows.append(L.fasp_hip_time_matrix(C.byref(A), 0, 20, C.byref(k)) * 1e3)
    L.fasp_hip_tune(b"kind", -1); L.fasp_hip_tune(b"lanes", -1)
    return (f"mxv with k_csr_xtile allowed {out[0]:7.1f} us, k_csr_wstream2 {out[1]:7.1f} us, k_csr_rows<4/8/16> "
            + "/".join(f"{v:.1f}" for v in rows))
for lev in levels:
    r, c, mi, mj, mv = H.matrix(lev, 0)
    A = sp.csr_matrix((mv.copy(), mj.copy(), mi.copy()), shape=(r, c))
    B = 12.0 * A.nnz + 20.0 * r
    print(f"{'var ' if var else ''}level {lev}: rows {r} nnz/row {A.nnz/r:.1f} plain bytes {B/1e9:.3f} GB", flush=True)
    print("   natural order :", t(A), flush=True)
    Ad, keep = T.as_csr(mi, mj, mv)
    for chunk in (65536, 262144):
        order = np.zeros(r, np.int32)
        t0 = time.time(); L.fasp_hip_cluster_order(C.byref(Ad), chunk, order.ctypes.data_as(T.c_int_p)); dt = time.time() - t0
        Ap = A[order][:, order]   # (scipy sorts nothing here: storage order of every row is kept)
        print(f"   cluster order (chunk {chunk}, {dt:.3f} s):", t(Ap), flush=True)
