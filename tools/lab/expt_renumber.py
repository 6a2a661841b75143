import os, sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
import faspsolver_amd as fa
from faspsolver_amd import _types as T
import bench as B
n = int(os.environ.get("BENCH_N", 256)); lev = int(os.environ.get("LEV", 2))
if os.environ.get("VAR"):
    ia, ja, a, f = fa.poisson7pt_var(n)
else:
    ia, ja, a, f, ue = fa.poisson7pt(n)
itp, amgp = B.workload_params()
H = fa.AMG(ia, ja, a, amgp, host_only=True)
r, c, mi, mj, mv = H.matrix(lev, 0)
A = sp.csr_matrix((mv.copy(), mj.copy(), mi.copy()), shape=(r, c))
H.close()
print("level", lev, "rows", r, "nnz", A.nnz, flush=True)
def run(M, tag):
    M = M.tocsr(); M.sort_indices() if False else None
    p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.max_levels = 2
    t = time.time()
    Hh = fa.AMG(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data, p)
    k = Hh.kernel_info(0, 0)[0]
    L = fa.lib()
    out = []
    for xt in (1, 0):
        L.fasp_hip_tune(b"xtile", xt)
        out.append((Hh.kernel_info(0, 0)[0], min(Hh.time_kernel(0, 0, 20) for _ in range(3)) * 1e3, min(Hh.time_kernel(2, 0, 20) for _ in range(3)) * 1e3))
    L.fasp_hip_tune(b"xtile", 1)
    print(tag, "setup %.1fs" % (time.time() - t), " ".join(f"[kind {k} mxv {m:.1f} us jacobi {j:.1f} us]" for k, m, j in out), flush=True)
    Hh.close()
run(A, "original order ")
t = time.time(); perm = reverse_cuthill_mckee(A, symmetric_mode=True); print("rcm %.1fs" % (time.time() - t), flush=True)
Ap = A[perm][:, perm]
run(Ap, "RCM order      ")
# brick order: BFS clusters of 64 (python would be slow) -> approximate by sorting RCM positions in blocks: skip
