"""tests/golden/p7_sweeps_256.npz: the sequential smoothers at the size of the metric, from the REFERENCE ITSELF
(oracle/_ref/libfasp_ref.so; build container only, ~10 minutes single-threaded): P7(256) with the reference's default
smoother (Gauss-Seidel, C/F order) and with SOR(1.1) in natural order -- iteration counts, residual histories, solution
samples.      python tools/gen_golden_sweeps_256.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G  # noqa: E402
from gen_golden_sweeps import MODS  # noqa: E402

if __name__ == "__main__":
    n = 256
    path = os.path.join(ROOT, "tests", "golden", "p7_sweeps_256.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}   # (tags already there are kept: a run adds what is missing)
    ia, ja, a, f, ue = G.ref_p7(n)
    for tag in ("gscf", "sor11", "gsnat"):   # (gsnat: round 5)
        if f"{tag}_iters" in out:
            continue
        st, xs, hist = G.solve(ia, ja, a, f, MODS[tag])
        out[f"{tag}_iters"] = np.array(st)
        out[f"{tag}_hist"] = hist
        out[f"{tag}_relres"] = np.array(hist[-1] / hist[0])
        step = max(1, len(xs) // 4096)
        out[f"{tag}_xsample"] = xs[::step].copy()
        print(n, tag, "iters", st, "relres %.10e" % (hist[-1] / hist[0]), flush=True)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "p7_sweeps_256.npz"), **out)
    print("wrote tests/golden/p7_sweeps_256.npz")
