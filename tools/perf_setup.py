import time, sys, os
import numpy as np
sys.path.insert(0, '.')
import faspsolver_amd as fa
import bench as B
n = int(os.environ.get("BENCH_N", 256))
t0 = time.perf_counter()
ia, ja, a, f, ue = fa.poisson7pt(n)
print("generate %.2f s" % (time.perf_counter() - t0), flush=True)
itp, amgp = B.workload_params()
for rep in range(2):
    t0 = time.perf_counter()
    H = fa.AMG(ia, ja, a, amgp)
    print("create %.2f s" % (time.perf_counter() - t0), flush=True)
    H.close()
