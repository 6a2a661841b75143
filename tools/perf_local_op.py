"""The level-0 / level-1 operators a rank holds of a row-partitioned P7(n) (rank r of nranks; host-only partition plan), timed through the
resident upload path on ONE GPU, beside the whole-level operator of the same row count: what a rank's SpMV costs on a multi-GPU run.
python tools/perf_local_op.py [n] [nranks] [rank]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rk = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667
H = fa.AMG(ia, ja, a, p, host_only=True)
H.dist_plan(rk, nr, 200000)
for lev in range(3):
    info = H.dist_info(lev)
    if info["replicated"]:
        break
    for which, name in ((0, "A"), (1, "P"), (2, "R")):
        m, c, lia, lja, lv = H.dist_matrix(lev, which)
        M, keep = T.as_csr(lia, lja, lv, ncol=c)
        for op, opn in ((0, "y = A x"), (2, "Jacobi")) if which == 0 else ((0, "y = A x"),):
            kind = C.c_int(-1)
            ms = L.fasp_hip_time_matrix(C.byref(M), op, 20, C.byref(kind))
            nnz = len(lv)
            print(f"level {lev} {name} rank {rk}/{nr}: {m} x {c}, {nnz} nnz, {opn}: {ms * 1e3:8.1f} us, kernel kind {kind.value}, plain-CSR bytes {(12 * nnz + 4 * m + 8 * c + 8 * m) / ms / 1e6:7.1f} GB/s", flush=True)
