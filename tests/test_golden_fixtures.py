"""The oracle AND the product's host-side logic against fixtures generated from the
reference itself (tools/gen_golden.py -> tests/golden/*.npz, reference compiled from its
own sources by oracle/Makefile).  Bit-exact: the setup is integer/graph work plus
floating-point expressions evaluated in the reference's order."""
import ctypes as C
import os

import numpy as np
import pytest

from _libs import DATA, OrcAMG, ROOT, T, default_params, oracle, orc_solve, poisson7pt, read_csr, read_vec

G = os.path.join(ROOT, "tests", "golden")


def _mods():
    def jac(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_JACOBI; a.relaxation = 0.6667
    def jacw(i, a): jac(i, a); a.cycle_type = T.W_CYCLE
    def jac22(i, a): jac(i, a); a.presmooth_iter = 2; a.postsmooth_iter = 2
    def l1(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_L1DIAG
    def gscf(i, a): i.tol = 1e-8
    def sor(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SOR; a.relaxation = 1.1
    def sgs(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SGS
    return {"jacobi_V": jac, "jacobi_W": jacw, "jacobi_V22": jac22, "l1diag_V": l1, "gscf_V": gscf,
            "sor_V": sor, "sgs_V": sgs}


def test_abi_layout_matches_reference():
    z = np.load(os.path.join(G, "abi.npz"))
    ours = [C.sizeof(T.dCSRmat), C.sizeof(T.dvector), C.sizeof(T.ITS_param), C.sizeof(T.AMG_param)]
    assert list(z["sizeof"][:4]) == ours == [40, 16, 40, 224]
    assert z["sizeof"][7] == C.sizeof(T.ivector) == 16
    offs = [T.AMG_param.tol.offset, T.AMG_param.coarse_dof.offset, T.AMG_param.relaxation.offset,
            T.AMG_param.amli_coef.offset, T.AMG_param.strong_threshold.offset, T.AMG_param.theta.offset,
            T.AMG_param.smoother.offset, T.AMG_param.ILU_levels.offset, T.AMG_param.SWZ_levels.offset]
    assert list(z["offsetof_amgparam"]) == offs


def test_param_defaults_match_reference(fa):
    z = np.load(os.path.join(G, "param_defaults.npz"))
    itp, amgp = default_params()
    assert bytes(itp) == z["its"].tobytes()
    assert bytes(amgp) == z["amg"].tobytes()
    assert bytes(fa.param_solver_init()) == z["its"].tobytes()
    assert bytes(fa.param_amg_init()) == z["amg"].tobytes()


def test_generator_matches_reference(fa):
    z = np.load(os.path.join(G, "p7_12.npz"))
    for gen in (poisson7pt, fa.poisson7pt):
        ia, ja, a, f, ue = gen(12)
        assert np.array_equal(ia, z["ia"]) and np.array_equal(ja, z["ja"])
        assert np.array_equal(a, z["a"]) and np.array_equal(f, z["f"]) and np.array_equal(ue, z["ue"])


def _check_hierarchy(z, get, nl):
    assert nl == int(z["num_levels"])
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            r, c, ia, ja, val = get(l, which)
            assert [r, c, len(val)] == list(z[f"L{l}_{nm}_shape"])
            assert np.array_equal(ia, z[f"L{l}_{nm}_ia"])
            assert np.array_equal(ja, z[f"L{l}_{nm}_ja"])
            assert np.array_equal(val, z[f"L{l}_{nm}_val"])  # bit-exact


def test_oracle_hierarchy_bit_exact():
    z = np.load(os.path.join(G, "p7_12.npz"))
    itp, amgp = default_params(); _mods()["jacobi_V"](itp, amgp)
    A, keep = T.as_csr(z["ia"], z["ja"], z["a"])
    H = OrcAMG(A, amgp)

    def get(l, which):
        m = [H.level(l).A, H.level(l).P, H.level(l).R][which]
        return (m.row, m.col) + T.csr_arrays(m)
    _check_hierarchy(z, get, H.num_levels)
    for l in range(H.num_levels - 1):
        assert np.array_equal(np.ctypeslib.as_array(H.level(l).cfmark.val, (H.level(l).A.row,)), z[f"L{l}_cfmark"])
    H.free()


def test_product_host_hierarchy_bit_exact(fa):
    z = np.load(os.path.join(G, "p7_12.npz"))
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    H = fa.AMG(z["ia"], z["ja"], z["a"], amgp, host_only=True)
    _check_hierarchy(z, H.matrix, H.num_levels)
    for l in range(H.num_levels - 1):
        assert np.array_equal(H.cfmark(l), z[f"L{l}_cfmark"])
    assert amgp.tentative_smooth == 1.0  # PreAMGSetupRS.c:83 mutation is reproduced
    H.close()


def test_oracle_kernels_bit_exact():
    z = np.load(os.path.join(G, "p7_12.npz"))
    O = oracle()
    A, keep = T.as_csr(z["ia"], z["ja"], z["a"])
    n = len(z["f"])
    x = z["k_x"].copy(); y = np.zeros(n)
    O.orc_mxv(C.byref(A), T.dp(x), T.dp(y))
    assert np.array_equal(y, z["k_mxv"])
    for alpha, nm in ((1.0, "p1"), (-1.0, "m1"), (0.7, "a07")):
        yy = z["k_y0"].copy()
        O.orc_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(yy))
        assert np.array_equal(yy, z[f"k_aAxpy_{nm}"])
    u = x.copy(); f = z["f"].copy()
    O.orc_smoother_jacobi(T.dp(u), 0, n - 1, 1, C.byref(A), T.dp(f), 1, 0.6667)
    assert np.array_equal(u, z["k_jacobi1"])
    y0 = z["k_y0"].copy()
    assert O.orc_dotprod(n, T.dp(x), T.dp(y0)) == float(z["k_dot"])


def test_oracle_precond_apply_bit_exact():
    z = np.load(os.path.join(G, "p7_12.npz"))
    itp, amgp = default_params(); _mods()["jacobi_V"](itp, amgp)
    A, keep = T.as_csr(z["ia"], z["ja"], z["a"])
    H = OrcAMG(A, amgp)
    r = z["pc_r"].copy(); out = np.zeros_like(r)
    oracle().orc_precond_amg(H.buf, C.byref(amgp), T.dp(r), T.dp(out))
    assert np.array_equal(out, z["pc_z"])
    H.free()


@pytest.mark.parametrize("name", list(_mods().keys()))
def test_oracle_histories_bit_exact(name):
    z = np.load(os.path.join(G, "p7_12.npz"))
    itp, amgp = default_params(); _mods()[name](itp, amgp)
    st, x, hist, rr = orc_solve(z["ia"], z["ja"], z["a"], z["f"], itp, amgp)
    assert st == int(z[f"solve_{name}_iters"])
    ref_hist = z[f"solve_{name}_hist"]  # r handed to the preconditioner (k < iters) + final true residual
    mine = np.concatenate([hist[:-2], hist[-1:]])
    assert np.array_equal(mine, ref_hist)
    assert np.array_equal(x, z[f"solve_{name}_x"])


def test_fe_fixture():
    z = np.load(os.path.join(G, "fe.npz"))
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    itp, amgp = default_params()
    A, keep = T.as_csr(ia, ja, a)
    H = OrcAMG(A, amgp)
    assert H.num_levels == int(z["num_levels"])
    for l in range(H.num_levels):
        Lv = H.level(l)
        assert [Lv.A.row, Lv.A.col, Lv.A.nnz] == list(z[f"L{l}_A_shape"])
        for nm in ("A", "P", "R"):
            if f"L{l}_{nm}_sum" in z:
                i2, j2, v2 = T.csr_arrays(getattr(Lv, nm))
                assert np.array_equal(np.array([v2.sum(), np.abs(v2).sum(), float(j2.astype(np.int64).sum())]),
                                      z[f"L{l}_{nm}_sum"])
        if f"L{l}_cfmark" in z:
            assert np.array_equal(np.ctypeslib.as_array(Lv.cfmark.val, (Lv.A.row,)), z[f"L{l}_cfmark"])
    H.free()
    for nm in ("jacobi_V", "gscf_V", "l1diag_V"):
        itp, amgp = default_params(); _mods()[nm](itp, amgp)
        st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
        assert st == int(z[f"solve_{nm}_iters"])
        assert np.array_equal(np.concatenate([hist[:-2], hist[-1:]]), z[f"solve_{nm}_hist"])


@pytest.mark.parametrize("n", [24, 40])
def test_midsize_summaries(n, fa):
    z = np.load(os.path.join(G, "p7_summaries.npz"))
    ia, ja, a, f, ue = poisson7pt(n)
    itp, amgp = default_params(); _mods()["jacobi_V"](itp, amgp)
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == int(z[f"n{n}_iters"])
    assert np.array_equal(np.concatenate([hist[:-2], hist[-1:]]), z[f"n{n}_hist"])
    p = fa.param_amg_init(); p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667
    H = fa.AMG(ia, ja, a, p, host_only=True)
    lv = [[H.matrix(l, 0)[0], len(H.matrix(l, 0)[4])] for l in range(H.num_levels)]
    assert lv == z[f"n{n}_levels"].tolist()
    H.close()


@pytest.mark.parametrize("tag", ["gscf", "gsnat", "sor11"])
def test_oracle_sweeps_bit_exact_64(tag):
    """The oracle's sequential smoothers at 64^3 against the reference's own run (tests/golden/p7_sweeps.npz,
    tools/gen_golden_sweeps.py): Gauss-Seidel in C/F order (the reference's defaults), in natural order, SOR(1.1) --
    bit for bit, as at the small sizes.  (The device parity test of the same fixture: tests/test_gpu_scale.py.)"""
    z = np.load(os.path.join(G, "p7_sweeps.npz"))
    ia, ja, a, f, ue = poisson7pt(64)
    itp, amgp = default_params(); itp.tol = 1e-8
    if tag == "gsnat": amgp.smooth_order = 0
    if tag == "sor11": amgp.smoother = T.SMOOTHER_SOR; amgp.relaxation = 1.1; amgp.smooth_order = 0
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == int(z[f"{tag}_iters"])
    assert np.array_equal(np.concatenate([hist[:-2], hist[-1:]]), z[f"{tag}_hist"])
    step = max(1, len(x) // 4096)
    assert np.array_equal(x[::step], z[f"{tag}_xsample"])


# --- F7: smoothed aggregation + GMRES family (config-5 shape, small) -----------------------
def _c5(i, a):
    i.tol = 1e-8; i.itsolver_type = 6; i.restart = 30
    a.AMG_type = T.SA_AMG; a.smoother = T.SMOOTHER_JACOBI; a.cycle_type = T.W_CYCLE


def _c5v(i, a):
    _c5(i, a); i.itsolver_type = 5; a.cycle_type = T.V_CYCLE


def test_sa_product_host_hierarchy_vs_golden(fa):
    z = np.load(os.path.join(G, "sa_p7_12.npz"))
    ia, ja, a, f, ue = poisson7pt(12)
    itp, amgp = default_params(); _c5(itp, amgp)
    H = fa.AMG(ia, ja, a, amgp, host_only=True)
    _check_hierarchy(z, H.matrix, H.num_levels)
    H.close()


@pytest.mark.parametrize("name,mod", [("vfgmres_W", _c5), ("vgmres_V", _c5v)])
def test_sa_gmres_oracle_vs_golden(name, mod):
    z = np.load(os.path.join(G, "sa_p7_12.npz"))
    ia, ja, a, f, ue = poisson7pt(12)
    itp, amgp = default_params(); mod(itp, amgp)
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == int(z[f"solve_{name}_iters"])
    assert np.array_equal(x, z[f"solve_{name}_x"])


# --- F6: block (BSR) path --------------------------------------------------------------------
def test_bsr_oracle_kernels_vs_golden():
    from _libs import read_bsr
    z = np.load(os.path.join(G, "bsr.npz"))
    ia, ja, val, nb = read_bsr(DATA + "/bsrmat_SPE01.dat")
    A, keep = T.as_bsr(ia, ja, val, nb)
    O = oracle()
    O.orc_bsr_mxv.argtypes = [C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    O.orc_bsr_getdiaginv.restype = T.c_double_p
    O.orc_bsr_getdiaginv.argtypes = [C.POINTER(T.dBSRmat)]
    x = z["spe01_x"].copy(); y = np.zeros(A.ROW * nb)
    O.orc_bsr_mxv(C.byref(A), T.dp(x), T.dp(y))
    assert np.array_equal(y, z["spe01_mxv"])
    d = np.ctypeslib.as_array(O.orc_bsr_getdiaginv(C.byref(A)), (A.ROW * nb * nb,))
    assert np.array_equal(d, z["spe01_diaginv"])


def _check_bsr_hierarchy(z, nl, get, diaginv):
    assert nl == int(z["p8_num_levels"])
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            ROW, COL, NNZ, ia, ja, val = get(l, which)
            assert [ROW, COL, NNZ] == list(z[f"p8_L{l}_{nm}_shape"])
            assert np.array_equal(ia, z[f"p8_L{l}_{nm}_ia"]) and np.array_equal(ja, z[f"p8_L{l}_{nm}_ja"])
            assert np.array_equal(val, z[f"p8_L{l}_{nm}_val"])
        if l < nl - 1:
            assert np.array_equal(diaginv(l), z[f"p8_L{l}_diaginv"])


def test_bsr_hierarchies_vs_golden(fa):
    from _libs import OrcBSR, bsr_params, poisson7pt_bsr
    z = np.load(os.path.join(G, "bsr.npz"))
    ia, ja, val, nb = poisson7pt_bsr(8)
    _, p1 = bsr_params(); _, p2 = bsr_params()
    H = OrcBSR(ia, ja, val, nb, p1)
    _check_bsr_hierarchy(z, H.num_levels, lambda l, w: H.levels[l]["APR"[w]], lambda l: H.levels[l]["diaginv"])
    Gp = fa.BSRAMG(ia, ja, val, nb, p2, host_only=True)
    _check_bsr_hierarchy(z, Gp.num_levels, Gp.matrix, Gp.diaginv)
    assert p1.strong_coupled == p2.strong_coupled == float(z["p8_strong_coupled_after"])
    Gp.free()


@pytest.mark.parametrize("name,solver,cycle", [("vgmres_V", 5, 1), ("pcg_V", 1, 1), ("vfgmres_W", 6, 2)])
def test_bsr_oracle_solves_vs_golden(name, solver, cycle):
    from _libs import bsr_params, orc_bsr_solve, poisson7pt_bsr
    z = np.load(os.path.join(G, "bsr.npz"))
    ia, ja, val, nb = poisson7pt_bsr(8)
    itp, amgp = bsr_params(solver, cycle)
    st, x, nl, rr = orc_bsr_solve(ia, ja, val, nb, z["p8_f"], itp, amgp)
    assert st == int(z[f"p8_{name}_iters"])
    assert np.array_equal(x, z[f"p8_{name}_x"])


def test_bsr_oracle_spe01_vs_golden():
    from _libs import bsr_params, orc_bsr_solve, read_bsr
    z = np.load(os.path.join(G, "bsr.npz"))
    ia, ja, val, nb = read_bsr(DATA + "/bsrmat_SPE01.dat"); f = read_vec(DATA + "/rhs_SPE01.dat")
    itp, amgp = bsr_params(); itp.maxit = 12
    st, x, nl, rr = orc_bsr_solve(ia, ja, val, nb, f, itp, amgp)
    assert st == int(z["spe01_status_maxit12"]) and nl == 1
    assert np.array_equal(x, z["spe01_x_maxit12"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,solver,cycle", [("vgmres_V", 5, 1), ("pcg_V", 1, 1), ("vfgmres_W", 6, 2)])
def test_gpu_bsr_solves_vs_golden(fa, name, solver, cycle):
    """The device block path against the REFERENCE's committed outputs (no oracle in between)."""
    from _libs import bsr_params, poisson7pt_bsr
    z = np.load(os.path.join(G, "bsr.npz"))
    ia, ja, val, nb = poisson7pt_bsr(8)
    itp, amgp = bsr_params(solver, cycle)
    x = np.zeros(len(z["p8_f"]))
    st = fa.solver_dbsr_krylov_amg(ia, ja, val, nb, z["p8_f"], x, itp, amgp)
    assert st == int(z[f"p8_{name}_iters"])
    xr = z[f"p8_{name}_x"]
    assert np.abs(x - xr).max() <= 1e-10 * np.abs(xr).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name,mod", [("vfgmres_W", _c5), ("vgmres_V", _c5v)])
def test_gpu_sa_gmres_vs_golden(fa, name, mod):
    z = np.load(os.path.join(G, "sa_p7_12.npz"))
    ia, ja, a, f, ue = poisson7pt(12)
    itp, amgp = default_params(); mod(itp, amgp)
    x = np.zeros(len(f))
    st = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, itp, amgp)
    assert st == int(z[f"solve_{name}_iters"])
    xr = z[f"solve_{name}_x"]
    assert np.abs(x - xr).max() <= 1e-10 * np.abs(xr).max()
