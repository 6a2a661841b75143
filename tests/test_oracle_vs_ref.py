"""Direct comparison of the oracle (and the product's host setup) with the reference
compiled from its own sources (oracle/_ref/libfasp_ref.so).  Skipped where that build is
absent and cannot be made (no /root/reference)."""
import ctypes as C

import numpy as np
import pytest

from _libs import (DATA, OrcAMG, T, default_params, have_ref, oracle, orc_solve, poisson7pt, read_csr,
                   read_vec, ref, ref_solve)

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def R():
    r = ref()
    if r is None:
        pytest.skip("oracle/_ref/libfasp_ref.so not available")
    return r


def _mods():
    def jac(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_JACOBI; a.relaxation = 0.6667
    def jacw(i, a): jac(i, a); a.cycle_type = T.W_CYCLE
    def vw(i, a): jac(i, a); a.cycle_type = T.VW_CYCLE
    def l1(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_L1DIAG
    def gscf(i, a): i.tol = 1e-8
    def gsn(i, a): i.tol = 1e-8; a.smooth_order = T.NO_ORDER
    def sor(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SOR; a.relaxation = 1.1
    def ssor(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SSOR; a.relaxation = 1.2
    def sgs(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SGS
    def gsor(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_GSOR; a.relaxation = 0.9; a.cycle_type = T.W_CYCLE
    def sgsor(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_SGSOR; a.relaxation = 1.05
    def theta(i, a): jac(i, a); a.strong_threshold = 0.6; a.truncation_threshold = 0.4
    def precres(i, a): jac(i, a); i.stop_type = T.STOP_REL_PRECRES
    def modres(i, a): jac(i, a); i.stop_type = T.STOP_MOD_REL_RES
    # STOP_MOD_REL_RES with x0 = 0 drives the coarse safe CG into ERROR_SOLVER_SOLSTAG: this case
    # exercises the SPVGMRES safety net of PreMGUtil.inl:50.
    def vg(i, a): jac(i, a); i.itsolver_type = T.SOLVER_VGMRES; i.restart = 30
    def vg4(i, a): jac(i, a); i.itsolver_type = T.SOLVER_VGMRES; i.restart = 4
    def vfg(i, a): jac(i, a); i.itsolver_type = T.SOLVER_VFGMRES; i.restart = 30
    def vfg5w(i, a): jac(i, a); i.itsolver_type = T.SOLVER_VFGMRES; i.restart = 5; a.cycle_type = T.W_CYCLE
    def poly3(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_POLY
    def poly5w(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_POLY; a.polynomial_degree = 5; a.cycle_type = T.W_CYCLE; a.presmooth_iter = 2
    def poly1(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_POLY; a.polynomial_degree = 1; i.maxit = 30   # degree 1: the correction stays zero
    def jacf(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_JACOBIF; a.relaxation = 0.8
    def jacf2(i, a): i.tol = 1e-8; a.smoother = T.SMOOTHER_JACOBIF; a.relaxation = 1.0; a.presmooth_iter = 2; a.postsmooth_iter = 3
    def stdint(i, a): jac(i, a); a.interpolation_type = 2
    def stdint_w(i, a): i.tol = 1e-8; a.interpolation_type = 2; a.cycle_type = T.W_CYCLE; a.truncation_threshold = 0.4
    def amli1(i, a): jac(i, a); a.cycle_type = T.AMLI_CYCLE; a.amli_degree = 1
    def amli2gs(i, a): i.tol = 1e-8; a.cycle_type = T.AMLI_CYCLE; a.amli_degree = 2
    def amli2cs(i, a): jac(i, a); a.cycle_type = T.AMLI_CYCLE; a.amli_degree = 2; a.coarse_scaling = 1
    def amli0(i, a): jac(i, a); a.cycle_type = T.AMLI_CYCLE; a.amli_degree = 0
    def amli3vg(i, a): vg(i, a); a.cycle_type = T.AMLI_CYCLE; a.amli_degree = 3
    def namli_gcg(i, a): jac(i, a); a.cycle_type = T.NL_AMLI_CYCLE; a.nl_amli_krylov_type = 7
    def namli_gcr_gs(i, a): i.tol = 1e-8; a.cycle_type = T.NL_AMLI_CYCLE; a.nl_amli_krylov_type = 8
    def namli_ua(i, a): jac(i, a); a.cycle_type = T.NL_AMLI_CYCLE; a.AMG_type = T.UA_AMG   # pairwise aggregation + K-cycle: Notay's method
    def namli_sa_vfg(i, a): vfg(i, a); a.cycle_type = T.NL_AMLI_CYCLE; a.AMG_type = T.SA_AMG
    def fmg(i, a): jac(i, a); i.precond_type = T.PREC_FMG
    def fmg_gs_cs(i, a): i.tol = 1e-8; i.precond_type = T.PREC_FMG; a.coarse_scaling = 1
    def fmg_sa_vfg(i, a): vfg(i, a); i.precond_type = T.PREC_FMG; a.AMG_type = T.SA_AMG
    def gsf2w(i, a): i.tol = 1e-8; a.smoother = 12; a.presmooth_iter = 2; a.postsmooth_iter = 2; a.cycle_type = T.W_CYCLE   # SMOOTHER_GSF
    def cgsm1(i, a): vfg(i, a); a.smoother = 4          # SMOOTHER_CG: a nonlinear preconditioner, hence flexible GMRES
    def cgsm3(i, a): vfg(i, a); a.smoother = 4; a.presmooth_iter = 3; a.postsmooth_iter = 3
    def ac2(i, a): jac(i, a); a.coarsening_type = T.COARSE_AC; a.aggressive_path = 2
    def ac2_std_w(i, a): vfg(i, a); a.coarsening_type = T.COARSE_AC; a.aggressive_path = 2; a.interpolation_type = 2; a.cycle_type = T.W_CYCLE; a.aggressive_level = 2
    def ac1(i, a): jac(i, a); a.coarsening_type = T.COARSE_AC
    def mis(i, a): jac(i, a); a.coarsening_type = T.COARSE_MIS
    def rdc_jacf(i, a): i.tol = 1e-8; a.interpolation_type = T.INTERP_RDC; a.smoother = T.SMOOTHER_JACOBIF   # reduction-based AMG: the setup sets the F-Jacobi weight
    def rdc_gs_w(i, a): i.tol = 1e-8; a.interpolation_type = T.INTERP_RDC; a.cycle_type = T.W_CYCLE
    def mis_ext_w(i, a): jac(i, a); a.coarsening_type = T.COARSE_MIS; a.interpolation_type = T.INTERP_EXT; a.cycle_type = T.W_CYCLE
    def vgpre(i, a): vg(i, a); i.stop_type = T.STOP_REL_PRECRES
    def vfgmod(i, a): vfg(i, a); i.stop_type = T.STOP_MOD_REL_RES
    return dict(jac=jac, jacw=jacw, vw=vw, l1=l1, gscf=gscf, gsn=gsn, sor=sor, ssor=ssor, sgs=sgs,
                theta=theta, precres=precres, modres=modres, gsor=gsor, sgsor=sgsor, vg=vg, vg4=vg4, vfg=vfg, vfg5w=vfg5w,
                vgpre=vgpre, vfgmod=vfgmod, poly3=poly3, poly5w=poly5w, poly1=poly1, jacf=jacf, jacf2=jacf2, stdint=stdint, stdint_w=stdint_w, amli1=amli1, amli2gs=amli2gs, amli2cs=amli2cs, amli0=amli0, amli3vg=amli3vg, namli_gcg=namli_gcg, namli_gcr_gs=namli_gcr_gs, namli_ua=namli_ua, namli_sa_vfg=namli_sa_vfg, fmg=fmg, fmg_gs_cs=fmg_gs_cs, fmg_sa_vfg=fmg_sa_vfg, gsf2w=gsf2w, cgsm1=cgsm1, cgsm3=cgsm3, ac2=ac2, ac2_std_w=ac2_std_w, ac1=ac1, mis=mis, mis_ext_w=mis_ext_w, rdc_jacf=rdc_jacf, rdc_gs_w=rdc_gs_w)


@pytest.mark.parametrize("name", list(_mods().keys()))
@pytest.mark.parametrize("n", [10, 20])
def test_histories_bit_exact(R, n, name):
    ia, ja, a, f, ue = poisson7pt(n)
    i1, a1 = default_params(); _mods()[name](i1, a1)
    i2, a2 = default_params(); _mods()[name](i2, a2)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1)
    s2, x2, h2 = ref_solve(ia, ja, a, f, i2, a2)
    assert s1 == s2
    assert np.array_equal(x1, x2)
    if name in ("jac", "jacw", "vw", "l1", "gscf", "gsn", "sor", "ssor", "sgs", "theta", "gsor", "sgsor", "poly3", "poly5w"):
        # PCG with STOP_REL_RES: the recorded preconditioner inputs are exactly the residuals
        assert np.array_equal(np.concatenate([h1[:-2], h1[-1:]]), h2)


def _matrix(name):
    if name == "fe":
        from _libs import DATA, read_csr
        return read_csr(DATA + "/csrmat_FE.dat")
    if name.startswith("pos"):   # P7 with 15 % of the couplings made positive: exercises rem_positive_ff
        ia, ja, a = poisson7pt(int(name[3:]))[:3]
        a = a.copy()
        rng = np.random.default_rng(2)
        off = ja != np.repeat(np.arange(len(ia) - 1), np.diff(ia))
        flip = off & (rng.random(len(a)) < 0.15)
        a[flip] = np.abs(a[flip]) * 0.9
        return ia, ja, a
    return poisson7pt(int(name))[:3]


@pytest.mark.parametrize("coarsening", [1, 2, 5], ids=["RS", "RSP", "MIS"])
@pytest.mark.parametrize("interp", [1, 2, 6, 4], ids=["direct", "standard", "extended", "reduction"])
@pytest.mark.parametrize("name", ["9", "28", "fe", "pos16"])
def test_hierarchy_bit_exact_oracle_and_product(R, fa, name, interp, coarsening):
    """Direct (PreAMGInterp.c:302) and standard (:547, pattern PreAMGCoarsenRS.c:2006) interpolation; classical
    splitting (cfsplitting_cls) and the variant that keeps positive couplings (COARSE_RSP, cfsplitting_clsp :806)."""
    ia, ja, a = _matrix(name)
    i1, a1 = default_params(); a1.smoother = T.SMOOTHER_JACOBI; a1.interpolation_type = interp; a1.coarsening_type = coarsening
    i2, a2 = default_params(); a2.smoother = T.SMOOTHER_JACOBI; a2.interpolation_type = interp; a2.coarsening_type = coarsening
    a3 = fa.param_amg_init(); a3.smoother = T.SMOOTHER_JACOBI; a3.interpolation_type = interp; a3.coarsening_type = coarsening
    A, keep = T.as_csr(ia, ja, a)
    O = OrcAMG(A, a1)
    hr = R.ref_amg_setup_rs(C.byref(A), C.byref(a2))
    P = fa.AMG(ia, ja, a, a3, host_only=True)
    nl = R.ref_amg_num_levels(hr)
    assert O.num_levels == nl == P.num_levels
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            v = T.dCSRmat(); R.ref_amg_get_matrix(hr, l, which, C.byref(v))
            ref_arr = T.csr_arrays(v)
            mine = T.csr_arrays(getattr(O.level(l), nm))
            prod = P.matrix(l, which)[2:]
            for x, y, z in zip(ref_arr, mine, prod):
                assert np.array_equal(x, y) and np.array_equal(x, z)
    R.ref_amg_free(hr, C.byref(a2)); O.free(); P.close()
    assert bytes(a1) == bytes(a2) == bytes(a3)  # the same mutations of AMG_param


@pytest.mark.parametrize("alvl", [0, 2], ids=["lvl-default", "lvl2"])
@pytest.mark.parametrize("path", [1, 2], ids=["path1", "path2"])
@pytest.mark.parametrize("interp", [1, 2], ids=["direct", "standard"])
@pytest.mark.parametrize("name", ["9", "24", "fe", "pos16"])
def test_aggressive_coarsening_hierarchy_bit_exact(R, fa, name, interp, path, alvl):
    """COARSE_AC (cfsplitting_agg, PreAMGCoarsenRS.c:1435, couplings between C points :1065 / :1243): standard
    interpolation is forced while it is active; on the level where it ends the reference fills the standard
    pattern with the user's interpolation (PreAMGInterp.c:68-71 after PreAMGSetupRS.c:198-199) -- the `direct`
    cases cover that."""
    ia, ja, a = _matrix(name)
    def prm(x):
        x.smoother = T.SMOOTHER_JACOBI; x.interpolation_type = interp; x.coarsening_type = T.COARSE_AC
        x.aggressive_path = path; x.aggressive_level = alvl
        return x
    a1 = prm(default_params()[1]); a2 = prm(default_params()[1]); a3 = prm(fa.param_amg_init())
    A, keep = T.as_csr(ia, ja, a)
    O = OrcAMG(A, a1)
    hr = R.ref_amg_setup_rs(C.byref(A), C.byref(a2))
    P = fa.AMG(ia, ja, a, a3, host_only=True)
    nl = R.ref_amg_num_levels(hr)
    assert O.num_levels == nl == P.num_levels
    for l in range(nl):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == nl - 1:
                continue
            v = T.dCSRmat(); R.ref_amg_get_matrix(hr, l, which, C.byref(v))
            ref_arr = T.csr_arrays(v)
            mine = T.csr_arrays(getattr(O.level(l), nm))
            prod = P.matrix(l, which)[2:]
            for x, y, z in zip(ref_arr, mine, prod):
                assert np.array_equal(x, y) and np.array_equal(x, z)
    R.ref_amg_free(hr, C.byref(a2)); O.free(); P.close()
    assert bytes(a1) == bytes(a2) == bytes(a3)


def test_coarse_spvgmres_bit_exact(R):
    ia, ja, a, f, ue = poisson7pt(14)
    i1, a1 = default_params(); a1.smoother = T.SMOOTHER_JACOBI
    A, keep = T.as_csr(ia, ja, a)
    O = OrcAMG(A, a1)
    Ac = O.level(O.num_levels - 1).A
    n = Ac.row
    b = np.random.default_rng(3).standard_normal(n)
    o = oracle()
    o.orc_gmres.argtypes = [C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_void_p,
                            C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    R.ref_coarse_spvgmres.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_double,
                                      C.c_int, C.c_int]
    for restart, maxit, tol in ((20, 1000, 1e-10), (5, 1000, 1e-10), (20, 7, 1e-10), (3, 50, 1e-14)):
        x1 = np.zeros(n); x2 = np.zeros(n)
        bv, bk = T.as_vec(b); xv1 = T.dvector(n, T.dp(x1)); xv2 = T.dvector(n, T.dp(x2))
        s1 = o.orc_gmres(2, C.byref(Ac), C.byref(bv), C.byref(xv1), None, None, tol, 0.0, maxit, restart, 1, 0, None)
        s2 = R.ref_coarse_spvgmres(C.byref(Ac), C.byref(bv), C.byref(xv2), tol, maxit, restart)
        assert s1 == s2 and np.array_equal(x1, x2)
    O.free()


def test_coarse_spcg_bit_exact(R):
    # the coarsest-level solver on a matrix of the kind it sees: a Galerkin coarse operator
    ia, ja, a, f, ue = poisson7pt(14)
    i1, a1 = default_params(); a1.smoother = T.SMOOTHER_JACOBI
    A, keep = T.as_csr(ia, ja, a)
    O = OrcAMG(A, a1)
    Ac = O.level(O.num_levels - 1).A
    n = Ac.row
    rng = np.random.default_rng(7)
    b = rng.standard_normal(n)
    x1 = np.zeros(n); x2 = np.zeros(n)
    bv, bk = T.as_vec(b); xv1 = T.dvector(n, T.dp(x1)); xv2 = T.dvector(n, T.dp(x2))
    s1 = oracle().orc_spcg(C.byref(Ac), C.byref(bv), C.byref(xv1), 1e-10, max(250, min(n * n, 1000)), 1, 0)
    s2 = R.ref_coarse_spcg(C.byref(Ac), C.byref(bv), C.byref(xv2), 1e-10)
    assert s1 == s2 and np.array_equal(x1, x2)
    O.free()


def test_standalone_blas_restatements_equal_reference(R):
    """orc_vmv / _mxv_agg / _aAxpy_agg / _array_ax / _axpyz / _norm1 against the compiled reference's functions of
    SURVEY.md section 8 rows a10-a12, bit for bit (serial build: the same left-to-right loops)."""
    O = oracle()
    rng = np.random.default_rng(11)
    n, m = 500, 431
    ia = [0]; ja = []
    for i in range(n):
        cols = rng.choice(m, size=int(rng.integers(0, 12)), replace=False)
        ja.extend(cols.tolist()); ia.append(len(ja))
    ia = np.array(ia, np.int32); ja = np.array(ja, np.int32); a = rng.standard_normal(len(ja))
    A, keep = T.as_csr(ia, ja, a, ncol=m)
    x = rng.standard_normal(m); yv = rng.standard_normal(n)
    P = C.POINTER
    for lib, pre in ((O, "orc_"), (R, "fasp_blas_dcsr_")):
        getattr(lib, pre + "vmv").restype = C.c_double
        getattr(lib, pre + "vmv").argtypes = [P(T.dCSRmat), T.c_double_p, T.c_double_p]
        getattr(lib, pre + "mxv_agg").argtypes = [P(T.dCSRmat), T.c_double_p, T.c_double_p]
        getattr(lib, pre + "aAxpy_agg").argtypes = [C.c_double, P(T.dCSRmat), T.c_double_p, T.c_double_p]
    assert O.orc_vmv(C.byref(A), T.dp(x), T.dp(yv)) == R.fasp_blas_dcsr_vmv(C.byref(A), T.dp(x), T.dp(yv))
    y1 = np.zeros(n); y2 = np.ones(n)
    O.orc_mxv_agg(C.byref(A), T.dp(x), T.dp(y1)); R.fasp_blas_dcsr_mxv_agg(C.byref(A), T.dp(x), T.dp(y2))
    assert np.array_equal(y1, y2)
    for alpha in (1.0, -1.0, 0.7):
        y1 = yv.copy(); y2 = yv.copy()
        O.orc_aAxpy_agg(alpha, C.byref(A), T.dp(x), T.dp(y1)); R.fasp_blas_dcsr_aAxpy_agg(alpha, C.byref(A), T.dp(x), T.dp(y2))
        assert np.array_equal(y1, y2)
    O.orc_array_ax.argtypes = [C.c_int, C.c_double, T.c_double_p]; R.fasp_blas_darray_ax.argtypes = O.orc_array_ax.argtypes
    O.orc_axpyz.argtypes = [C.c_int, C.c_double, T.c_double_p, T.c_double_p, T.c_double_p]
    R.fasp_blas_darray_axpyz.argtypes = O.orc_axpyz.argtypes
    O.orc_norm1.restype = C.c_double; O.orc_norm1.argtypes = [C.c_int, T.c_double_p]
    R.fasp_blas_darray_norm1.restype = C.c_double; R.fasp_blas_darray_norm1.argtypes = O.orc_norm1.argtypes
    for s in (1.0, -0.3):
        x1 = x.copy(); x2 = x.copy()
        O.orc_array_ax(m, s, T.dp(x1)); R.fasp_blas_darray_ax(m, s, T.dp(x2))
        assert np.array_equal(x1, x2)
    xs = x[:n].copy() if m >= n else np.resize(x, n)
    z1 = np.zeros(n); z2 = np.ones(n)
    O.orc_axpyz(n, 0.37, T.dp(xs), T.dp(yv), T.dp(z1)); R.fasp_blas_darray_axpyz(n, 0.37, T.dp(xs), T.dp(yv), T.dp(z2))
    assert np.array_equal(z1, z2)
    assert O.orc_norm1(m, T.dp(x)) == R.fasp_blas_darray_norm1(m, T.dp(x))
