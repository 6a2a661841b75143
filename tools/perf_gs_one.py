"""One level, one sweep kind of the parity-mode Gauss-Seidel sweeps (seq_split.hip.h), timed alone -- with a library built with
EXTRA_HIPFLAGS=-DFLOW_TIMING the dataflow kernel prints where its compute waves spend their cycles.
python tools/perf_gs_one.py n level kind(10 ascending, 11 descending, 12 C rows, 13 F rows) [tune=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
n, lev, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    fa.lib().fasp_hip_tune(k.encode(), int(v))
ia, ja, a, f, ue = fa.poisson7pt(n)
H = fa.AMG(ia, ja, a, fa.param_amg_init())
H.set_rhs(f)
print(f"level {lev} kind {kind}: {H.time_kernel(kind, lev, 3) * 1e3:.1f} us per sweep", flush=True)
