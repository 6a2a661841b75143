"""tests/golden/p7_scale.npz, the n256_* keys: the HEADLINE configuration at the metric's own size from the REFERENCE
ITSELF (oracle/_ref/libfasp_ref.so: fasp_solver_dcsr_krylov_amg, SolCSR.c:476, with the Jacobi smoother of
ItrSmootherCSR.c:98; build container only, about five minutes single-threaded and 45 GB): P7(256), classical AMG
V(1,1), w-Jacobi 0.6667, PCG to 1e-8 -- iteration count, the whole residual history, a 4096-entry sample of the
solution and its sums.  Every other key of the fixture is kept as it is (tools/gen_golden_f5.py writes those).

    python tools/gen_golden_f5_256.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G  # noqa: E402

if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "p7_scale.npz")
    out = dict(np.load(path))
    n = 256
    ia, ja, a, f, ue = G.ref_p7(n)
    st, xs, hist = G.solve(ia, ja, a, f, G.MODS["jacobi_V"])
    assert st == int(out["n256_iters"]), (st, out["n256_iters"])
    out["n256_hist"] = hist
    out["n256_relres"] = np.array(hist[-1] / hist[0])
    out["n256_xsum"] = np.array([xs.sum(), np.abs(xs).max(), np.sqrt((xs * xs).sum())])
    step = max(1, len(xs) // 4096)
    out["n256_xsample"] = xs[::step].copy()
    print(n, "iters", st, "relres %.10e" % (hist[-1] / hist[0]), flush=True)
    np.savez_compressed(path, **out)
    print("wrote", path)
