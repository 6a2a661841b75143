"""Configs 3 and 5 of BASELINE.json at full size on one GPU, with oracle parity at a size the CPU
finishes in seconds (dev tool; output kept under profiles/).

  config 3: P7(n) (x) B3 block system, UA-AMG (VMB) + block Jacobi + VGMRES(30), tol 1e-8   [n = 128]
  config 5: Q1 27-point anisotropic diffusion (1, 1, 0.01), SA-AMG + W-cycle + VFGMRES(30)    [n = 123]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
import _libs

which = sys.argv[1] if len(sys.argv) > 1 else "3"
n_full = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if which == "3" else 123)
n_par = int(sys.argv[3]) if len(sys.argv) > 3 else (48 if which == "3" else 40)


C5_SMOOTHER = os.environ.get("C5_SMOOTHER", "jacobi")  # "jacobi": w-Jacobi(0.6667) as config 2; "default": the reference's GS


def c5(itp, amgp):
    itp.tol = 1e-8; itp.itsolver_type = 6; itp.restart = 30
    amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE
    if C5_SMOOTHER == "jacobi":
        amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667


def run3(n, parity):
    ia, ja, a, f0, ue = fa.poisson7pt(n)
    nb = 3
    val = (a[:, None, None] * _libs.B3[None, :, :]).reshape(-1)
    f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
    itp, amgp = _libs.bsr_params(5)
    t0 = time.time()
    G = fa.BSRAMG(ia, ja, val, nb, amgp)
    ts = time.time() - t0
    for rep in range(2):
        st, x, hist, stats = G.solve(f, itp)
    print(f"config 3  P7({n})xB3: block rows {len(ia)-1} DOF {len(f)} NNZ {len(ja)} levels {G.num_levels} "
          f"setup+upload {ts:.2f} s | iters {st} relres {stats.relres:.6e} solve {stats.solve_seconds*1e3:.1f} ms "
          f"= {len(f)/stats.solve_seconds:.3e} DOF/s  cycles {stats.vcycles} coarse GMRES its {stats.coarse_iters}", flush=True)
    if parity:
        i2, a2 = _libs.bsr_params(5)
        t0 = time.time()
        s1, x1, nl, rr = _libs.orc_bsr_solve(ia, ja, val, nb, f, i2, a2)
        print(f"   oracle (1 core): iters {s1} relres {rr:.6e} {time.time()-t0:.1f} s   "
              f"max|dx|/max|x| {np.abs(x-x1).max()/np.abs(x1).max():.2e}", flush=True)
    G.free()


def run5(n, parity):
    ia, ja, a, f = fa.aniso27pt(n)
    itp, amgp = fa.param_solver_init(), fa.param_amg_init()
    c5(itp, amgp)
    t0 = time.time()
    H = fa.AMG(ia, ja, a, amgp)
    ts = time.time() - t0
    H.set_rhs(f)
    for rep in range(2):
        st, hist, stats = H.solve_resident(itp)
    x = H.get_solution()
    print(f"config 5 [{C5_SMOOTHER}]  aniso27pt({n}): rows {len(f)} nnz {len(a)} levels {H.num_levels} setup+upload {ts:.2f} s | "
          f"iters {st} relres {stats.relres:.6e} solve {stats.solve_seconds*1e3:.1f} ms = "
          f"{len(f)/stats.solve_seconds:.3e} DOF/s  cycles {stats.vcycles} coarse its {stats.coarse_iters}", flush=True)
    if parity:
        i2, a2 = _libs.default_params(); c5(i2, a2)
        t0 = time.time()
        s1, x1, h1, rr = _libs.orc_solve(ia, ja, a, f, i2, a2)
        print(f"   oracle: iters {s1} relres {rr:.6e} {time.time()-t0:.1f} s   max|dx|/max|x| "
              f"{np.abs(x-x1).max()/np.abs(x1).max():.2e}", flush=True)
    H.close()


run = run3 if which == "3" else run5
run(n_par, True)
run(n_full, False)
