"""A few GS-smoothed solves of P7(n) with the reference's defaults (for rocprofv3: python3 tools/gs_solve.py [n] [solves])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
H = fa.AMG(ia, ja, a, fa.param_amg_init()); H.set_rhs(f)
for r in range(reps):
    st, hist, stats = H.solve_resident(itp)
    print(f"solve {r}: {st} iterations, relres {stats.relres:.10e}, {stats.solve_seconds*1e3:.1f} ms", flush=True)
