"""Development experiment: k_csr_xtile against k_csr_wstream2 on a 27-point operator (n^3 grid), rows in lexicographic
order (64 consecutive rows = a grid line: 2.9 entries per distinct column) and in 4 x 4 x 4 bricks (8 per distinct column)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
I = sp.identity(n, format="csr")
T1 = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
S = sp.kron(sp.kron(T1, T1), T1, format="csr")          # 27-point pattern
rng = np.random.default_rng(3)
S.data = -rng.uniform(0.5, 1.5, S.nnz)
d = np.asarray(-S.sum(axis=1)).ravel() + 1.0
A = (S + sp.diags(d - S.diagonal())).tocsr()
print("rows", A.shape[0], "nnz", A.nnz, flush=True)
def run(M, tag):
    M = M.tocsr()
    p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.max_levels = 2
    Hh = fa.AMG(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data, p)
    L = fa.lib()
    out = []
    for xt in (1, 0, 1, 0):
        L.fasp_hip_tune(b"xtile", xt)
        out.append((Hh.kernel_info(0, 0)[0], min(Hh.time_kernel(0, 0, 20) for _ in range(3)) * 1e3, min(Hh.time_kernel(2, 0, 20) for _ in range(3)) * 1e3))
    L.fasp_hip_tune(b"xtile", 1)
    plain = 12.0 * M.nnz + 20.0 * M.shape[0]
    print(tag, " ".join(f"[kind {k}: mxv {m:.1f} us = {plain/m/1e6:.0f} GB/s plain-CSR equivalent, jacobi {j:.1f} us]" for k, m, j in out), flush=True)
    Hh.close()
run(A, "lexicographic:")
idx = np.arange(n ** 3)
z, y, x = idx // (n * n), (idx // n) % n, idx % n
key = (((z // 4) * (n // 4) + (y // 4)) * (n // 4) + (x // 4)) * 64 + ((z % 4) * 4 + (y % 4)) * 4 + (x % 4)
perm = np.argsort(key)
run(A[perm][:, perm], "4x4x4 bricks: ")
