"""Variable-coefficient P7(n) solve alone (rocprofv3 target / development tool): python tools/perf_var.py [n] [solves]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
import bench as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ia, ja, a, f = fa.poisson7pt_var(n)
itp, amgp = B.workload_params()
t = time.time(); H = fa.AMG(ia, ja, a, amgp); print(f"setup+upload {time.time()-t:.2f} s, levels {H.num_levels}", flush=True)
H.set_rhs(f)
for r in range(reps):
    st, hist, stats = H.solve_resident(itp)
    print(f"solve: iters {st} relres {stats.relres:.10e} {stats.solve_seconds*1e3:.2f} ms coarse its {stats.coarse_iters}", flush=True)
for l in range(H.num_levels):
    print("level", l, [H.kernel_info(l, w)[0] for w in ((0, 1, 2) if l < H.num_levels - 1 else (0,))], flush=True)
H.close()
