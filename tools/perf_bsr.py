"""BSR path on the config-3 shapes (dev tool): SpMV bandwidth on SPE01 and P7(n) (x) B3, then the
resident fasp_hip_bsr_solve (UA-AMG + VGMRES(30), tol 1e-8) on P7(n) (x) B3."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
import _libs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = fa.lib()


def synthetic(n):
    ia, ja, a, f, ue = fa.poisson7pt(n)
    val = (a[:, None, None] * _libs.B3[None, :, :]).reshape(-1)
    return ia, ja, val, 3


cases = [("SPE01",) + tuple(_libs.read_bsr(_libs.DATA + "/bsrmat_SPE01.dat")), (f"P7({n})xB3",) + synthetic(n)]
for name, ia, ja, val, nb in cases:
    A, keep = T.as_bsr(ia, ja, val, nb)
    ms = L.fasp_hip_time_bsr_mxv(C.byref(A), 20)
    B = A.NNZ * (8 * nb * nb + 4) + 4 * (A.ROW + 1) + 16 * A.ROW * nb
    print(f"{name}: ROW {A.ROW} NNZ {A.NNZ} nb {nb}: {ms*1e3:.1f} us/launch, {B/ms/1e6:.0f} GB/s algorithmic "
          f"({B/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)

name, ia, ja, val, nb = cases[1]
f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
for solver, label in ((5, "VGMRES(30)"), (1, "PCG")):
    itp, amgp = _libs.bsr_params(solver)
    t0 = time.time()
    G = fa.BSRAMG(ia, ja, val, nb, amgp)
    t_setup = time.time() - t0
    st, x, hist, stats = G.solve(f, itp)
    st, x, hist, stats = G.solve(f, itp)
    dof = len(f)
    print(f"{name} {label}: levels {G.num_levels} setup {t_setup:.2f} s iters {st} relres {stats.relres:.3e} "
          f"solve {stats.solve_seconds*1e3:.1f} ms ({dof/stats.solve_seconds:.3e} DOF/s) cycles {stats.vcycles} "
          f"coarse its {stats.coarse_iters}", flush=True)
    if n <= 32:
        i2, a2 = _libs.bsr_params(solver)
        t0 = time.time()
        s1, x1, nl, rr = _libs.orc_bsr_solve(ia, ja, val, nb, f, i2, a2)
        print(f"   oracle: iters {s1} relres {rr:.3e} {time.time()-t0:.2f} s  max|dx|/max|x| "
              f"{np.abs(x-x1).max()/np.abs(x1).max():.2e}", flush=True)
    G.free()
