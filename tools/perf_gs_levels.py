"""Per-level cost of the parity-mode Gauss-Seidel sweeps (the reference's sequential order; seq_split.hip.h): microseconds per
sweep of every level of P7(n) for the four schedules (all rows ascending / descending, C rows, F rows), next to the Jacobi sweep
of the same level.  FASP_HIP_SETUP_TIMING=1 prints the schedules.  python tools/perf_gs_levels.py [n] [tune=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    fa.lib().fasp_hip_tune(k.encode(), int(v))
ia, ja, a, f, ue = fa.poisson7pt(n)
p = fa.param_amg_init()
H = fa.AMG(ia, ja, a, p)
H.set_rhs(f)
tot = [0.0] * 5
for lev in range(H.num_levels - 1):
    t = [H.time_kernel(k, lev, 5) * 1e3 for k in (2, 10, 11, 12, 13)]
    for i, v in enumerate(t): tot[i] += v
    r = H.matrix(lev, 0)[0]
    print(f"level {lev} rows {r:8d}: Jacobi {t[0]:8.1f} us | GS ascending {t[1]:9.1f}  descending {t[2]:9.1f}  C rows {t[3]:9.1f}  F rows {t[4]:9.1f}", flush=True)
print(f"sum {' '.join(sys.argv[2:]):16s} : Jacobi {tot[0]:8.1f} us | GS ascending {tot[1]:9.1f}  descending {tot[2]:9.1f}  C rows {tot[3]:9.1f}  F rows {tot[4]:9.1f}")
