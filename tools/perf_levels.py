"""Per-level kernel timings on the resident hierarchy (development tool).

usage: python tools/perf_levels.py [n] [reps]
Prints, for every level of the P7(n) hierarchy, the mean duration of each kernel class
(HIP events on the launch stream) and the algorithmic bandwidth it corresponds to
(SURVEY.md section 8d byte formulas).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa  # noqa: E402
from faspsolver_amd import _types as T  # noqa: E402


def spmv_bytes(row, col, nnz):
    return 12 * nnz + 4 * (row + 1) + 8 * col + 8 * row


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    t = time.time()
    ia, ja, a, f, ue = fa.poisson7pt(n)
    print(f"P7({n}): rows {len(f)} nnz {len(a)} gen {time.time()-t:.2f}s", flush=True)
    amgp = fa.param_amg_init()
    amgp.smoother = T.SMOOTHER_JACOBI
    amgp.relaxation = 0.6667
    amgp.print_level = 2
    t = time.time()
    H = fa.AMG(ia, ja, a, amgp)
    print(f"setup+upload {time.time()-t:.2f}s levels {H.num_levels}", flush=True)
    names = {0: "mxv", 1: "aAxpy-1", 2: "jacobi", 5: "mxv+dot", 6: "R mxv", 7: "P aAxpy", 3: "dot", 4: "axpy"}
    tot = 0.0
    for l in range(H.num_levels):
        r, c, _, _, v = H.matrix(l, 0)
        nnz = len(v)
        line = f"L{l} rows {r:9d} nnz {nnz:10d} ({nnz/r:7.1f}/row):"
        for k in (0, 2, 5):
            ms = H.time_kernel(k, l, reps)
            gb = (spmv_bytes(r, c, nnz) + (8 * r if k == 2 else 0)) / ms / 1e6
            line += f" {names[k]} {ms*1e3:8.1f}us {gb:7.0f}GB/s |"
        if l < H.num_levels - 1:
            for k, w in ((6, 2), (7, 1)):
                rr, cc, _, _, vv = H.matrix(l, w)
                ms = H.time_kernel(k, l, reps)
                gb = spmv_bytes(rr, cc, len(vv)) / ms / 1e6
                line += f" {names[k]} {ms*1e3:8.1f}us {gb:7.0f}GB/s |"
        for k in (3, 4):
            ms = H.time_kernel(k, l, reps)
            gb = (16 if k == 3 else 24) * r / ms / 1e6
            line += f" {names[k]} {ms*1e3:7.1f}us {gb:6.0f}GB/s |"
        print(line, flush=True)
    itp = fa.param_solver_init()
    itp.tol = 1e-8
    itp.print_level = 2
    for rep in range(3):
        st, x, hist, stats = H.solve(f, itp)
        itp.print_level = 0
        print(f"solve: iters {st} relres {stats.relres:.10e} t {stats.solve_seconds*1e3:.2f} ms "
              f"spmv {stats.spmv_ms*1e3:.1f} us coarse_iters {stats.coarse_iters} vcycles {stats.vcycles} "
              f"DOF/s {len(f)/stats.solve_seconds:.3e}", flush=True)
    H.close()


if __name__ == "__main__":
    main()
