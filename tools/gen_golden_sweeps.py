"""Sequential-smoother fixtures at scale: tests/golden/p7_sweeps.npz from the REFERENCE ITSELF (oracle/_ref/libfasp_ref.so).
Run in the build container only; the fixture is data (iteration counts, residual histories, solution samples).

    python tools/gen_golden_sweeps.py

P7(64) and P7(128) with the reference's default smoother (Gauss-Seidel, C/F order), Gauss-Seidel in natural order and
SOR(1.1).  64^3: the deep levels run as one-workgroup triangular solves; 128^3 (p7_sweeps_128.npz): levels 1-2 have the wide
dependency classes that take the cluster form, the far entries and the helper workgroups (csrc/seq_split.hip.h).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G  # noqa: E402  (reference bindings; importing does not regenerate anything)

T = G.T
MODS = {
    "gscf": lambda i, a: (setattr(i, "tol", 1e-8),),
    "gsnat": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smooth_order", 0)),
    "sor11": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_SOR), setattr(a, "relaxation", 1.1),
                           setattr(a, "smooth_order", 0)),
}

if __name__ == "__main__":
    for n, name in ((64, "p7_sweeps.npz"), (128, "p7_sweeps_128.npz")):
        out = {}
        ia, ja, a, f, ue = G.ref_p7(n)
        for tag, mod in MODS.items():
            st, xs, hist = G.solve(ia, ja, a, f, mod)
            out[f"{tag}_iters"] = np.array(st)
            out[f"{tag}_hist"] = hist
            out[f"{tag}_relres"] = np.array(hist[-1] / hist[0])
            step = max(1, len(xs) // 4096)
            out[f"{tag}_xsample"] = xs[::step].copy()
            print(n, tag, "iters", st, "relres %.10e" % (hist[-1] / hist[0]), flush=True)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **out)
        print("wrote tests/golden/" + name)
