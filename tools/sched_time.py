"""First-solve time of a GS-smoothed hierarchy (= sweep schedule building): python tools/sched_time.py n [nat] [tune=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
n = int(sys.argv[1])
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
amgp = fa.param_amg_init()
for kv in sys.argv[2:]:
    if kv == "nat": amgp.smooth_order = 0
    else:
        k, v = kv.split("="); fa.lib().fasp_hip_tune(k.encode(), int(v))
H = fa.AMG(ia, ja, a, amgp); H.set_rhs(f)
for rep in range(2):
    st, hist, stats = H.solve_resident(itp)
    print("solve", rep, st, stats.relres, f"{stats.solve_seconds*1e3:.1f} ms", flush=True)
H2 = fa.AMG(ia, ja, a, amgp); H2.set_rhs(f)   # a second hierarchy in the same process: kernels loaded, what is left is the schedules
for rep in range(2):
    st, hist, stats = H2.solve_resident(itp)
    print("second hierarchy, solve", rep, st, f"{stats.solve_seconds*1e3:.1f} ms", flush=True)
