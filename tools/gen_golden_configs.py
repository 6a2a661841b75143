"""Full-size pins of BASELINE.json configs 3 and 5 from the REFERENCE ITSELF (oracle/_ref/libfasp_ref.so).

    python tools/gen_golden_configs.py [3|5|both]        (build container only; minutes of serial CPU time)

  config 3: P7(128) (x) B3 block system (2 097 152 block rows, nb = 3), UA-AMG (VMB) + block Jacobi + VGMRES(30),
            tol 1e-8, rhs = numpy default_rng(1).standard_normal
  config 5: Q1 27-point anisotropic diffusion (1, 1, 0.01) at n = 123 (1 860 867 rows, 49.4 M nnz),
            SA-AMG + W-cycle + w-Jacobi(0.6667) + VFGMRES(30), tol 1e-8

Output: tests/golden/configs_full.npz -- iteration counts, true relative residuals of the reference's solutions,
residual history (config 5), solution checksums and a strided sample of x.  Data only, no reference source.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import faspsolver_amd as fa  # noqa: E402
from faspsolver_amd import _types as T  # noqa: E402
import _libs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "configs_full.npz")


def csr_residual(ia, ja, a, x, f):
    import scipy.sparse as sp
    A = sp.csr_matrix((a, ja, ia), shape=(len(f), len(f)))
    r = f - A @ x
    return float(np.sqrt(r @ r) / np.sqrt(f @ f))


def xsummary(tag, x, out):
    out[f"{tag}_xsum"] = np.array([x.sum(), np.abs(x).max(), np.sqrt((x * x).sum())])
    step = max(1, len(x) // 4096)
    out[f"{tag}_xsample"] = x[::step].copy()


def config3(out, n=128):
    ia, ja, a, f0, ue = _libs.poisson7pt(n)
    nb = 3
    val = (a[:, None, None] * _libs.B3[None, :, :]).reshape(-1)
    f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
    itp, amgp = _libs.bsr_params(5)
    t0 = time.time()
    st, x = _libs.ref_bsr_solve(ia, ja, val, nb, f, itp, amgp)
    dt = time.time() - t0
    # true residual through the expanded scalar matrix (kron structure): r = f - (A (x) B3) x
    import scipy.sparse as sp
    A = sp.csr_matrix((a, ja, ia), shape=(len(ia) - 1, len(ia) - 1))
    X = x.reshape(-1, nb)
    r = f.reshape(-1, nb) - (A @ X) @ _libs.B3.T
    rr = float(np.sqrt((r * r).sum()) / np.sqrt(f @ f))
    out["c3_n"] = np.array(n); out["c3_iters"] = np.array(st); out["c3_relres_true"] = np.array(rr)
    xsummary("c3", x, out)
    print(f"config 3 P7({n})xB3: reference iters {st}, true relres {rr:.10e}, {dt:.1f} s", flush=True)


def c5(itp, amgp):
    itp.tol = 1e-8; itp.itsolver_type = 6; itp.restart = 30
    amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE
    amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667


def config5(out, n=123):
    ia, ja, a, f = fa.aniso27pt(n)
    itp, amgp = _libs.default_params(); c5(itp, amgp)
    t0 = time.time()
    st, x, hist = _libs.ref_solve(ia, ja, a, f, itp, amgp)
    dt = time.time() - t0
    rr = csr_residual(ia, ja, a, x, f)
    out["c5_n"] = np.array(n); out["c5_iters"] = np.array(st); out["c5_relres_true"] = np.array(rr)
    out["c5_hist"] = hist
    out["c5_shape"] = np.array([len(f), len(a)])
    xsummary("c5", x, out)
    print(f"config 5 aniso27pt({n}): reference iters {st}, true relres {rr:.10e}, hist end "
          f"{(hist[-1] / hist[0]) if len(hist) else float('nan'):.10e}, {dt:.1f} s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    if which in ("3", "both"):
        config3(out, int(os.environ.get("C3_N", "128")))
        np.savez_compressed(OUT, **out)
    if which in ("5", "both"):
        config5(out, int(os.environ.get("C5_N", "123")))
        np.savez_compressed(OUT, **out)
    print("wrote", OUT)
