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
print(f"level {lev} kind {kind}: {H.time_kernel(kind, lev, int(os.environ.get('REPS', '3'))) * 1e3:.1f} us per sweep", flush=True)
L = fa.lib()
if hasattr(L, "fasp_hip_flow_times"):   # FLOW_TIMING build: when every strip of the last launch started and ended
    import ctypes as C, numpy as np
    t = np.zeros(8192, np.uint64)
    L.fasp_hip_flow_times(t.ctypes.data_as(C.c_void_p), 8192)
    t = t.reshape(-1, 2); m = t[:, 1] > 0; t = t[m].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    print("strip: start .. end (us)")
    for s in range(0, len(t), max(1, len(t) // 48)):
        print(f"  {s:4d}: {t[s, 0] - t0:8.1f} .. {t[s, 1] - t0:8.1f}  ({t[s, 1] - t[s, 0]:6.1f})")
    print(f"  last end {t[:, 1].max() - t0:.1f} us, mean strip time {np.mean(t[:, 1] - t[:, 0]):.1f} us, {len(t)} strips")
