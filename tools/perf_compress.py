"""Dictionary-coded kernels (k_csr_dict8) vs the plain CSR kernels on the same resident hierarchy:
per-level timings and a bit-level comparison of the full solve (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
t0 = time.time()
H = fa.AMG(ia, ja, a, amgp)
print(f"P7({n}) setup+upload {time.time()-t0:.2f}s levels {H.num_levels}", flush=True)
itp = fa.param_solver_init(); itp.tol = 1e-8
names = {0: "mxv", 2: "jacobi", 5: "mxv+dot", 6: "R mxv", 7: "P aAxpy"}
res = {}
for comp, rpl, lds in ((0, -1, 1), (1, 1, 1), (1, 2, 1), (1, 1, 0), (1, 2, 0)):
    L.fasp_hip_tune(b"compress", comp)
    L.fasp_hip_tune(b"rpl", rpl)
    L.fasp_hip_tune(b"lds_tab", lds)
    print(f"--- compress = {comp} rpl = {rpl} lds_tab = {lds}")
    for l in range(min(H.num_levels, 4)):
        r, c, _, _, v = H.matrix(l, 0)
        line = f"L{l} rows {r:9d} nnz {len(v):10d}:"
        for k in (0, 2, 5) + ((6, 7) if l < H.num_levels - 1 else ()):
            ms = H.time_kernel(k, l, reps)
            line += f" {names[k]} {ms*1e3:8.1f}us |"
        print(line, flush=True)
    for rep in range(3):
        st, x, hist, stats = H.solve(f, itp)
    print(f"solve: iters {st} relres {stats.relres:.10e} t {stats.solve_seconds*1e3:.2f} ms spmv {stats.spmv_ms*1e3:.1f} us "
          f"DOF/s {len(f)/stats.solve_seconds:.3e}", flush=True)
    res[comp] = (st, x.copy(), hist.copy())
L.fasp_hip_tune(b"rpl", -1); L.fasp_hip_tune(b"lds_tab", 1)
print("iterations equal:", res[0][0] == res[1][0], " x bit-identical:", np.array_equal(res[0][1], res[1][1]),
      " history bit-identical:", np.array_equal(res[0][2], res[1][2]))
H.close()
