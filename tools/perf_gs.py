"""Reference DEFAULT parameters (GS smoother with C/F ordering) and SOR on P7(n): solve time of the parity mode (the
reference's sequential sweep as a parallel pass + a triangular solve, seq_split.hip.h; seq_flow 0 = one launch per
dependency class instead of the dataflow form) and of the flagged multicolour mode (fasp_hip_tune("gs_multicolor", 1): one launch per
colour, a different iteration), next to the Jacobi-smoothed solve of the benchmark.
Development tool: python tools/perf_gs.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ia, ja, a, f, ue = fa.poisson7pt(n)
itp = fa.param_solver_init(); itp.tol = 1e-8
L = fa.lib()
cases = (("Jacobi w=0.6667 (benchmark parameters)", lambda p: (setattr(p, "smoother", T.SMOOTHER_JACOBI), setattr(p, "relaxation", 0.6667))),
         ("GS-CF (reference defaults)", lambda p: None),
         ("GS natural order", lambda p: setattr(p, "smooth_order", 0)),
         ("SOR w=1.1, natural order", lambda p: (setattr(p, "smoother", T.SMOOTHER_SOR), setattr(p, "relaxation", 1.1), setattr(p, "smooth_order", 0))))
for name, mod in cases:
    amgp = fa.param_amg_init(); mod(amgp)
    t = time.time(); H = fa.AMG(ia, ja, a, amgp); print(f"P7({n}) {name}: setup {time.time()-t:.1f}s levels {H.num_levels}", flush=True)
    H.set_rhs(f)
    for mc, sb in (((0, 1), (0, 1), (0, 1)) if "Jacobi" in name else ((0, 1), (0, 1), (0, 1), (0, 0), (0, 0), (1, 1), (1, 1))):
        L.fasp_hip_tune(b"gs_multicolor", mc); L.fasp_hip_tune(b"seq_flow", sb)
        st, hist, stats = H.solve_resident(itp)
        print(f"  gs_multicolor {mc} seq_flow {sb}: iters {st} relres {stats.relres:.10e} solve {stats.solve_seconds*1e3:.1f} ms coarse its {stats.coarse_iters}", flush=True)
    H.close()
L.fasp_hip_tune(b"gs_multicolor", 0); L.fasp_hip_tune(b"seq_flow", 1)
