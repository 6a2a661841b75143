"""Development: one sweep kind of one level of P7(n) with the dataflow solve's launch capped at g workgroups (fasp_hip_tune seq_grid).
python tools/grid_sweep.py n level kind(10 ascending, 11 descending, 12 C rows, 13 F rows)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
n = int(sys.argv[1]); lev = int(sys.argv[2]); kind = int(sys.argv[3])
L = fa.lib(); L.fasp_hip_tune(b"seq_spine", 0)
ia, ja, a, f, ue = fa.poisson7pt(n)
H = fa.AMG(ia, ja, a, fa.param_amg_init()); H.set_rhs(f)
for g in (0, -1, 1, 2, 4, 8, 16, 32, 64):   # 0: the schedule's own cap, -1: every resident workgroup
    L.fasp_hip_tune(b"seq_grid", g)
    print(f"level {lev} kind {kind} grid {g}: {H.time_kernel(kind, lev, 3) * 1e3:.1f} us", flush=True)
