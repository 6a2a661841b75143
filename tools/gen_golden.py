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
        cf = R.ref_amg_get_cfmark(h, l) if l < nl - 1 else None
        if cf:  # aggregation hierarchies carry no C/F marker
            out[f"L{l}_cfmark"] = np.ctypeslib.as_array(cf, (out[f"L{l}_A_shape"][0],)).copy()
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

    # F7: config-5 shape at small size: SA hierarchy + VFGMRES(30)/W and VGMRES/V on P7(12)
    ia, ja, a, f, ue = ref_p7(12)
    sa = {}
    def c5(i, a_):
        i.tol = 1e-8; i.itsolver_type = 6; i.restart = 30
        a_.AMG_type = T.SA_AMG; a_.smoother = T.SMOOTHER_JACOBI; a_.cycle_type = T.W_CYCLE
    def c5v(i, a_):
        c5(i, a_); i.itsolver_type = 5; a_.cycle_type = T.V_CYCLE
    itp, amgp = ref_params(); c5(itp, amgp)
    h, hier = hierarchy(ia, ja, a, amgp)
    R.ref_amg_free(h, C.byref(amgp))
    sa.update({k: v for k, v in hier.items() if "cfmark" not in k})
    for nm, mod in (("vfgmres_W", c5), ("vgmres_V", c5v)):
        st, xs, hist = solve(ia, ja, a, f, mod)
        sa[f"solve_{nm}_iters"] = np.array(st); sa[f"solve_{nm}_hist"] = hist; sa[f"solve_{nm}_x"] = xs
    np.savez_compressed(os.path.join(OUT, "sa_p7_12.npz"), **sa)

    # F6: block path.  SPE01 (shipped): SpMV, inverse diagonal blocks, the non-converging
    # one-level VGMRES run; synthetic P7(8) (x) B3: hierarchy + three solves
    from _libs import bsr_arrays, bsr_params, bsr_protos, poisson7pt_bsr, read_bsr, ref_bsr_solve
    bsr_protos()
    bs = {}
    ia, ja, val, nb = read_bsr(DATA + "/bsrmat_SPE01.dat"); f = read_vec(DATA + "/rhs_SPE01.dat")
    A, keep = T.as_bsr(ia, ja, val, nb)
    x = np.random.default_rng(7).standard_normal(A.COL * nb); y = np.zeros(A.ROW * nb)
    R.fasp_blas_dbsr_mxv.argtypes = [C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    R.fasp_blas_dbsr_mxv(C.byref(A), T.dp(x), T.dp(y))
    bs["spe01_x"] = x; bs["spe01_mxv"] = y
    R.fasp_dbsr_getdiaginv.restype = T.dvector; R.fasp_dbsr_getdiaginv.argtypes = [C.POINTER(T.dBSRmat)]
    d = R.fasp_dbsr_getdiaginv(C.byref(A))
    bs["spe01_diaginv"] = np.ctypeslib.as_array(d.val, (d.row,)).copy()
    itp, amgp = bsr_params(); itp.maxit = 12
    st, xs = ref_bsr_solve(ia, ja, val, nb, f, itp, amgp)
    bs["spe01_status_maxit12"] = np.array(st); bs["spe01_x_maxit12"] = xs
    n = 8
    ia, ja, val, nb = poisson7pt_bsr(n)
    f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
    bs["p8_f"] = f
    A, keep = T.as_bsr(ia, ja, val, nb)
    _, amgp = bsr_params()
    hb = R.ref_bsr_setup_ua(C.byref(A), C.byref(amgp))
    nl = R.ref_bsr_num_levels(hb)
    bs["p8_num_levels"] = np.array(nl); bs["p8_strong_coupled_after"] = np.array(amgp.strong_coupled)
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            v = T.dBSRmat(); R.ref_bsr_get_matrix(hb, l, which, C.byref(v))
            i2, j2, v2 = bsr_arrays(v)
            bs[f"p8_L{l}_{nm}_shape"] = np.array([v.ROW, v.COL, v.NNZ])
            bs[f"p8_L{l}_{nm}_ia"] = i2; bs[f"p8_L{l}_{nm}_ja"] = j2; bs[f"p8_L{l}_{nm}_val"] = v2
        if l < nl - 1:
            bs[f"p8_L{l}_diaginv"] = np.ctypeslib.as_array(
                R.ref_bsr_get_diaginv(hb, l), (int(bs[f"p8_L{l}_A_shape"][0]) * nb * nb,)).copy()
    R.ref_bsr_free(hb)
    for nm, (solver, cycle) in (("vgmres_V", (5, 1)), ("pcg_V", (1, 1)), ("vfgmres_W", (6, 2))):
        itp, amgp = bsr_params(solver, cycle)
        st, xs = ref_bsr_solve(ia, ja, val, nb, f, itp, amgp)
        bs[f"p8_{nm}_iters"] = np.array(st); bs[f"p8_{nm}_x"] = xs
    np.savez_compressed(os.path.join(OUT, "bsr.npz"), **bs)
    # F8b: parameter structs the reference builds from its own ini files (test/ini/*.dat, copied as data
    # fixtures to tests/golden/data/ini); AMG_param.polynomial_degree is masked: the reference's input
    # defaults leave it unset (AuxParam.c:100)
    import glob
    R.ref_param_from_file.argtypes = [C.c_char_p, C.POINTER(T.ITS_param), C.POINTER(T.AMG_param)]
    ini = {}
    for fn in sorted(glob.glob(os.path.join(DATA, "ini", "*.dat"))):
        itp, amgp = T.ITS_param(), T.AMG_param()
        R.ref_param_from_file(fn.encode(), C.byref(itp), C.byref(amgp))
        b = bytearray(bytes(amgp)); o = T.AMG_param.polynomial_degree.offset; b[o:o + 2] = b"\0\0"
        ini[os.path.basename(fn) + "_its"] = np.frombuffer(bytes(itp), np.uint8)
        ini[os.path.basename(fn) + "_amg"] = np.frombuffer(bytes(b), np.uint8)
    np.savez(os.path.join(OUT, "ini_params.npz"), **ini)

    for fn in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, fn)
        if os.path.isfile(p):
            print(f"{fn:24s} {os.path.getsize(p):9d} bytes")


if __name__ == "__main__":
    main()
