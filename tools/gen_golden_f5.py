"""F5 fixtures at scale: tests/golden/p7_scale.npz from the REFERENCE ITSELF
(oracle/_ref/libfasp_ref.so).  Run in the build container only; the fixture is data (level sizes,
iteration counts, residual histories, solution checksums), no reference source.

    python tools/gen_golden_f5.py

n = 64, 128: P7(n), jacobi_V (the headline configuration).  n = 48, 96: the variable-coefficient twin
(faspsolver_amd.poisson7pt_var).  n = 256: the oracle values recorded in BASELINE.md section 2 (the
reference needs 190 s and 45 GB for it; measured once while surveying).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G  # noqa: E402  (reference bindings; importing does not regenerate anything)
import faspsolver_amd as fa  # noqa: E402

R, T = G.R, G.T


def summary(tag, ia, ja, a, f, out):
    itp, amgp = G.ref_params(); G.MODS["jacobi_V"](itp, amgp)
    h, hier = G.hierarchy(ia, ja, a, amgp)
    R.ref_amg_free(h, C.byref(amgp))
    nl = int(hier["num_levels"])
    out[f"{tag}_levels"] = np.array([[hier[f"L{l}_A_shape"][0], hier[f"L{l}_A_shape"][2]] for l in range(nl)])
    del hier
    st, xs, hist = G.solve(ia, ja, a, f, G.MODS["jacobi_V"])
    out[f"{tag}_iters"] = np.array(st)
    out[f"{tag}_hist"] = hist
    out[f"{tag}_relres"] = np.array(hist[-1] / hist[0])
    out[f"{tag}_xsum"] = np.array([xs.sum(), np.abs(xs).max(), np.sqrt((xs * xs).sum())])
    step = max(1, len(xs) // 4096)
    out[f"{tag}_xsample"] = xs[::step].copy()
    print(tag, "levels", nl, "iters", st, "relres %.10e" % (hist[-1] / hist[0]), flush=True)


if __name__ == "__main__":
    out = {}
    for n in (64, 128):
        ia, ja, a, f, ue = G.ref_p7(n)
        summary(f"n{n}", ia, ja, a, f, out)
    for n in (48, 96):
        ia, ja, a, f = fa.poisson7pt_var(n, G.ref_p7(n))
        summary(f"var{n}", ia, ja, a, f, out)
    # BASELINE.md section 2 (oracle measurement of the unmodified serial reference)
    out["n256_iters"] = np.array(14)
    out["n256_relres"] = np.array(6.3426837114e-09)
    out["n256_levels"] = np.array([[16777216, 117047296], [8388608, 158205440], [1430459, 49708321], [257139, 16563015],
                                   [80619, 15431065], [44833, 17462821], [27390, 19448232], [16621, 18509031],
                                   [9459, 12823505], [4971, 6409103]])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "p7_scale.npz"), **out)
    print("wrote tests/golden/p7_scale.npz")


def abi_precond():
    """F8 extension: every field offset of AMG_data / precond_data in the compiled reference (oracle/ref_shim.c)."""
    np.savez(os.path.join(ROOT, "tests", "golden", "abi_precond.npz"),
             amgdata=np.array([R.ref_offsetof_amgdata(i) for i in range(27)]),
             precdata=np.array([R.ref_offsetof_precdata(i) for i in range(28)]))


if __name__ == "__main__":
    abi_precond()
