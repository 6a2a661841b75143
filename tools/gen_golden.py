"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libfasp_ref.so,
compiled by oracle/Makefile from /root/reference's own sources).  Run in the build
container only (the reference does not travel to the GPU box); the fixtures it writes are
committed.  Fixtures are data: inputs and expected outputs, no reference source.

    python tools/gen_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import DATA, T, read_csr, read_vec, ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
R = ref()
assert R is not None, "oracle/_ref/libfasp_ref.so missing: run `make -C oracle` with /root/reference present"


def ref_params():
    itp = T.ITS_param(); amgp = T.AMG_param()
    R.fasp_param_solver_init(C.byref(itp)); R.fasp_param_amg_init(C.byref(amgp))
    return itp, amgp


def ref_p7(n):
    A = T.dCSRmat(); b = T.dvector(); u = T.dvector()
    R.ref_poisson7pt(n, n, n, C.byref(A), C.byref(b), C.byref(u))
    ia, ja, a = T.csr_arrays(A)
    return ia, ja, a, np.ctypeslib.as_array(b.val, (b.row,)).copy(), np.ctypeslib.as_array(u.val, (u.row,)).copy()


def hierarchy(ia, ja, a, amgp):
    A, keep = T.as_csr(ia, ja, a)
    h = R.ref_amg_setup_rs(C.byref(A), C.byref(amgp))
    nl = R.ref_amg_num_levels(h)
    out = {"num_levels": np.array(nl)}
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            v = T.dCSRmat(); R.ref_amg_get_matrix(h, l, which, C.byref(v))
            i2, j2, v2 = T.csr_arrays(v)
            out[f"L{l}_{nm}_shape"] = np.array([v.row, v.col, v.nnz])
            out[f"L{l}_{nm}_ia"] = i2; out[f"L{l}_{nm}_ja"] = j2; out[f"L{l}_{nm}_val"] = v2
        if l < nl - 1:
            out[f"L{l}_cfmark"] = np.ctypeslib.as_array(R.ref_amg_get_cfmark(h, l), (out[f"L{l}_A_shape"][0],)).copy()
    return h, out


def solve(ia, ja, a, f, mod):
    itp, amgp = ref_params(); mod(itp, amgp)
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv, x = T.as_vec(x)
    hist = np.zeros(600); nh = C.c_int(0)
    st = R.ref_krylov_amg_hist(C.byref(A), C.byref(bv), C.byref(xv), C.byref(itp), C.byref(amgp),
                               T.dp(hist), 600, C.byref(nh))
    return st, x, hist[:nh.value].copy()


MODS = {
    "jacobi_V": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_JACOBI), setattr(a, "relaxation", 0.6667)),
    "jacobi_W": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_JACOBI), setattr(a, "relaxation", 0.6667), setattr(a, "cycle_type", T.W_CYCLE)),
    "jacobi_V22": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_JACOBI), setattr(a, "relaxation", 0.6667), setattr(a, "presmooth_iter", 2), setattr(a, "postsmooth_iter", 2)),
    "l1diag_V": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_L1DIAG)),
    "gscf_V": lambda i, a: (setattr(i, "tol", 1e-8),),
    "sor_V": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_SOR), setattr(a, "relaxation", 1.1)),
    "sgs_V": lambda i, a: (setattr(i, "tol", 1e-8), setattr(a, "smoother", T.SMOOTHER_SGS)),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    # F8: ABI facts
    np.savez(os.path.join(OUT, "abi.npz"),
             sizeof=np.array([R.ref_sizeof(i) for i in range(8)]),
             offsetof_amgparam=np.array([R.ref_offsetof_amgparam(i) for i in range(9)]))
    itp, amgp = ref_params()
    np.savez(os.path.join(OUT, "param_defaults.npz"), its=np.frombuffer(bytes(itp), np.uint8),
             amg=np.frombuffer(bytes(amgp), np.uint8))

    # F3/F4: P7(12): generator output, full hierarchy, kernels, precond apply, histories
    n = 12
    ia, ja, a, f, ue = ref_p7(n)
    fx = {"n": np.array(n), "ia": ia, "ja": ja, "a": a, "f": f, "ue": ue}
    itp, amgp = ref_params(); MODS["jacobi_V"](itp, amgp)
    h, hier = hierarchy(ia, ja, a, amgp)
    fx.update(hier)
    rng = np.random.default_rng(2024)
    x = rng.standard_normal(len(f)); y0 = rng.standard_normal(len(f))
    A, keep = T.as_csr(ia, ja, a)
    y = np.zeros(len(f)); R.fasp_blas_dcsr_mxv(C.byref(A), T.dp(x), T.dp(y))
    fx["k_x"] = x; fx["k_y0"] = y0; fx["k_mxv"] = y
    for alpha, nm in ((1.0, "p1"), (-1.0, "m1"), (0.7, "a07")):
        yy = y0.copy()
        R.fasp_blas_dcsr_aAxpy.argtypes = [C.c_double, C.POINTER(T.dCSRmat), T.c_double_p, T.c_double_p]
        R.fasp_blas_dcsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(yy))
        fx[f"k_aAxpy_{nm}"] = yy
    R.fasp_smoother_dcsr_jacobi.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int,
                                            C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int, C.c_double]
    u = x.copy(); uv = T.dvector(len(f), T.dp(u)); bv = T.dvector(len(f), T.dp(f))
    R.fasp_smoother_dcsr_jacobi(C.byref(uv), 0, len(f) - 1, 1, C.byref(A), C.byref(bv), 1, 0.6667)
    fx["k_jacobi1"] = u
    R.fasp_blas_darray_dotprod.argtypes = [C.c_int, T.c_double_p, T.c_double_p]
    fx["k_dot"] = np.array(R.fasp_blas_darray_dotprod(len(f), T.dp(x), T.dp(y0)))
    z = np.zeros(len(f)); r = rng.standard_normal(len(f))
    R.ref_precond_amg(h, C.byref(amgp), T.dp(r), T.dp(z))
    fx["pc_r"] = r; fx["pc_z"] = z
    R.ref_amg_free(h, C.byref(amgp))
    for nm, mod in MODS.items():
        st, xs, hist = solve(ia, ja, a, f, mod)
        fx[f"solve_{nm}_iters"] = np.array(st); fx[f"solve_{nm}_hist"] = hist; fx[f"solve_{nm}_x"] = xs
    np.savez_compressed(os.path.join(OUT, "p7_12.npz"), **fx)

    # F2: csrmat_FE (input shipped under tests/golden/data): level summary + histories
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    fe = {}
    itp, amgp = ref_params()
    h, hier = hierarchy(ia, ja, a, amgp)
    R.ref_amg_free(h, C.byref(amgp))
    nl = int(hier["num_levels"])
    fe["num_levels"] = hier["num_levels"]
    for l in range(nl):
        fe[f"L{l}_A_shape"] = hier[f"L{l}_A_shape"]
        # checksums instead of the arrays (the matrix is 27 k nnz per level)
        for nm in ("A", "P", "R"):
            if f"L{l}_{nm}_val" in hier:
                fe[f"L{l}_{nm}_sum"] = np.array([hier[f"L{l}_{nm}_val"].sum(), np.abs(hier[f"L{l}_{nm}_val"]).sum(),
                                                 float(hier[f"L{l}_{nm}_ja"].astype(np.int64).sum())])
        if f"L{l}_cfmark" in hier:
            fe[f"L{l}_cfmark"] = hier[f"L{l}_cfmark"].astype(np.int8)
    for nm in ("jacobi_V", "gscf_V", "l1diag_V"):
        st, xs, hist = solve(ia, ja, a, f, MODS[nm])
        fe[f"solve_{nm}_iters"] = np.array(st); fe[f"solve_{nm}_hist"] = hist
    reg = lambda i, a_: (setattr(i, "tol", 1e-10),)
    st, xs, hist = solve(ia, ja, a, f, reg)
    fe["solve_reg_iters"] = np.array(st); fe["solve_reg_hist"] = hist
    np.savez_compressed(os.path.join(OUT, "fe.npz"), **fe)

    # F5: summaries for mid sizes (rows / nnz per level, iterations, history)
    summ = {}
    for n in (24, 40):
        ia, ja, a, f, ue = ref_p7(n)
        itp, amgp = ref_params(); MODS["jacobi_V"](itp, amgp)
        h, hier = hierarchy(ia, ja, a, amgp)
        R.ref_amg_free(h, C.byref(amgp))
        nl = int(hier["num_levels"])
        summ[f"n{n}_levels"] = np.array([[hier[f"L{l}_A_shape"][0], hier[f"L{l}_A_shape"][2]] for l in range(nl)])
        st, xs, hist = solve(ia, ja, a, f, MODS["jacobi_V"])
        summ[f"n{n}_iters"] = np.array(st); summ[f"n{n}_hist"] = hist
    np.savez_compressed(os.path.join(OUT, "p7_summaries.npz"), **summ)
    for fn in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, fn)
        if os.path.isfile(p):
            print(f"{fn:24s} {os.path.getsize(p):9d} bytes")


if __name__ == "__main__":
    main()
