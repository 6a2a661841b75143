"""First solve against later ones with the reference's default smoother (GS, C/F order) at 256^3: the first one builds the sweep
schedules on the host (FASP_HIP_SETUP_TIMING=1 prints them).  Development tool: python tools/perf_gs_first_solve.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
ia, ja, a, f, ue = fa.poisson7pt(256)
itp = fa.param_solver_init(); itp.tol = 1e-8
amgp = fa.param_amg_init()
H = fa.AMG(ia, ja, a, amgp); H.set_rhs(f)
for rep in range(2):
    t = time.time(); st, hist, stats = H.solve_resident(itp); print("solve", rep, st, "%.1f ms wall %.2f s" % (stats.solve_seconds*1e3, time.time()-t), flush=True)
