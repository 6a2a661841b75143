"""Config 4 size on ONE GPU: P7(512) (134 M DOF, 938 M nnz): host setup time, GPU solve,
and parity against the oracle run on the host cores with the same hierarchy."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
import bench as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.perf_counter()
ia, ja, a, f, ue = fa.poisson7pt(n)
print(f"P7({n}) rows {len(f)} nnz {len(a)} gen {time.perf_counter()-t0:.1f}s", flush=True)
itp, amgp = B.workload_params()
amgp.print_level = 2
t0 = time.perf_counter()
H = fa.AMG(ia, ja, a, amgp)
print(f"setup+upload {time.perf_counter()-t0:.1f}s levels {H.num_levels}", flush=True)
H.set_rhs(f)
for rep in range(3):
    st, hist, stats = H.solve_resident(itp)
    kind, mbytes = H.kernel_info(0, 0)     # bytes the level-0 kernel of the solve has to move (its stored matrix form + x + y)
    Bs = mbytes + 16.0 * len(f)
    print(f"GPU solve: iters {st} relres {stats.relres:.10e} t {stats.solve_seconds*1e3:.1f} ms  DOF/s {len(f)/stats.solve_seconds:.3e} "
          f"level-0 SpMV (kernel family {kind}) {stats.spmv_ms*1e3:.1f} us = {Bs/stats.spmv_ms/1e6:.0f} GB/s moved ({Bs/stats.spmv_ms/1e6/8000:.3f} of peak) coarse its {stats.coarse_iters}", flush=True)
x = H.get_solution()
print("max|x-u_exact|", np.max(np.abs(x - ue)), flush=True)
cb, its_cpu, rr_cpu, hist_dev = B.cpu_baseline(H, ia, ja, a, f, int(st), hist, float(os.environ.get("BENCH_CPU_BUDGET_S", "400")), int(os.environ.get("BENCH_CPU_THREADS", "64")))
print(json.dumps({"cpu": cb, "iters_cpu": its_cpu, "relres_cpu": rr_cpu, "relres_gpu": stats.relres, "hist_dev": hist_dev}), flush=True)
H.close()
