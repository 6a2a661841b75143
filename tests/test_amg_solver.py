"""AMG as a stand-alone solver: fasp_solver_amg (SolAMG.c:49) -> fasp_amg_solve (PreMGSolve.c:49).

Known answers: the reference's own tutorial/out/poisson-amg-c.out (csrmat_FE, defaults, maxit 50:
hierarchy sizes and the 4-cycle residual history as printed), the compiled reference (bit-exact x),
and on the GPU the oracle's full-precision history."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA, default_params, have_ref, oracle, poisson7pt, read_csr, read_vec, ref

needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def orc_amg_solve(ia, ja, a, f, p, x0=None, cap=300):
    o = oracle()
    o.orc_solver_amg.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector),
                                 C.POINTER(T.AMG_param), T.c_double_p, C.c_int, T.c_int_p, T.c_double_p]
    A, keep = T.as_csr(ia, ja, a)
    n = len(f)
    x = np.zeros(n) if x0 is None else x0.copy()
    bv, fk = T.as_vec(f); xv = T.dvector(n, T.dp(x))
    hist = np.zeros(cap); nh = C.c_int(0); rr = C.c_double(0)
    st = o.orc_solver_amg(C.byref(A), C.byref(bv), C.byref(xv), C.byref(p), T.dp(hist), cap, C.byref(nh),
                          C.byref(rr))
    return st, x, hist[:nh.value].copy(), rr.value


def ref_amg_solve(ia, ja, a, f, p, x0=None):
    R = ref()
    R.fasp_solver_amg.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector),
                                  C.POINTER(T.AMG_param)]
    A, keep = T.as_csr(ia, ja, a)
    n = len(f)
    x = np.zeros(n) if x0 is None else x0.copy()
    bv, fk = T.as_vec(f); xv = T.dvector(n, T.dp(x))
    return R.fasp_solver_amg(C.byref(A), C.byref(bv), C.byref(xv), C.byref(p)), x


def jac(p): p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667; p.tol = 1e-8; p.maxit = 100
def dflt(p): p.tol = 1e-8; p.maxit = 100
def wcyc(p): jac(p); p.cycle_type = T.W_CYCLE
def sa(p): jac(p); p.AMG_type = T.SA_AMG
def amli(p): jac(p); p.cycle_type = T.AMLI_CYCLE; p.amli_degree = 2   # fasp_amg_solve_amli, PreMGSolve.c:142
def namli(p): jac(p); p.cycle_type = T.NL_AMLI_CYCLE   # fasp_amg_solve_namli, PreMGSolve.c:230
def few(p): jac(p); p.maxit = 3
def sor(p): p.smoother = T.SMOOTHER_SOR; p.relaxation = 1.1; p.tol = 1e-8; p.maxit = 100


MODS = {"jacobi": jac, "default_gscf": dflt, "W": wcyc, "sa": sa, "maxit3": few, "sor": sor, "amli2": amli, "namli": namli}


def params(mod):
    _, p = default_params()
    mod(p)
    return p


def test_oracle_vs_reference_tutorial_output():
    """tutorial/out/poisson-amg-c.out of the reference: csrmat_FE, defaults, maxit 50, tol 1e-6."""
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    _, p = default_params(); p.maxit = 50
    st, x, hist, rr = orc_amg_solve(ia, ja, a, f, p)
    assert st == 4
    printed = [7.514358e+00, 7.403129e-02, 2.635624e-03, 1.325550e-04, 6.628261e-06]  # ||r|| column
    assert len(hist) == 5
    for h, q in zip(hist, printed):
        assert float("%.6e" % h) == q
    assert "%.6e" % rr == "8.820794e-07"


@needs_ref
@pytest.mark.parametrize("name", list(MODS))
def test_oracle_equals_reference(name):
    ia, ja, a, f, ue = poisson7pt(12)
    s1, x1, hist, rr = orc_amg_solve(ia, ja, a, f, params(MODS[name]))
    s2, x2 = ref_amg_solve(ia, ja, a, f, params(MODS[name]))
    assert s1 == s2
    assert np.array_equal(x1, x2)


@needs_ref
def test_oracle_equals_reference_guess_and_zero_rhs():
    ia, ja, a, f, ue = poisson7pt(12)
    x0 = np.random.default_rng(3).standard_normal(len(f))
    s1, x1, hist, rr = orc_amg_solve(ia, ja, a, f, params(jac), x0)
    s2, x2 = ref_amg_solve(ia, ja, a, f, params(jac), x0)
    assert s1 == s2 and np.array_equal(x1, x2)
    z = np.zeros(len(f))
    s1, x1, hist, rr = orc_amg_solve(ia, ja, a, z, params(jac), np.ones(len(f)))
    s2, x2 = ref_amg_solve(ia, ja, a, z, params(jac), np.ones(len(f)))
    assert s1 == s2 == 1 and np.array_equal(x1, x2) and not x1.any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MODS))
@pytest.mark.parametrize("n", [12, 24])
def test_gpu_amg_solver_matches_oracle(name, n):
    ia, ja, a, f, ue = poisson7pt(n)
    s1, x1, h1, rr1 = orc_amg_solve(ia, ja, a, f, params(MODS[name]))
    p = params(MODS[name])
    H = fa.AMG(ia, ja, a, p)
    s2, x2, h2, stats = H.amg_solve(f, p)
    assert s2 == s1
    assert len(h1) == len(h2)
    # residual histories agree to 1e-8 relative (+ rounding floor), solution to 1e-10
    assert np.allclose(h2, h1, rtol=1e-8, atol=1e-12 * h1[0])
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()
    H.close()


@pytest.mark.gpu
def test_gpu_amg_solver_dropin_with_guess_and_zero_rhs():
    ia, ja, a, f, ue = poisson7pt(16)
    x0 = np.random.default_rng(5).standard_normal(len(f))
    s1, x1, h1, rr1 = orc_amg_solve(ia, ja, a, f, params(jac), x0)
    x2 = x0.copy()
    s2 = fa.solver_amg(ia, ja, a, f, x2, params(jac))
    assert s1 == s2
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()
    x3 = np.ones(len(f))
    assert fa.solver_amg(ia, ja, a, np.zeros(len(f)), x3, params(jac)) == 1
    assert not x3.any()


@pytest.mark.gpu
def test_gpu_amg_solver_tutorial_case():
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    _, p = default_params(); p.maxit = 50
    x = np.zeros(len(f))
    assert fa.solver_amg(ia, ja, a, f, x, p) == 4
    A = np.zeros(0)
    r = f.copy()
    for i in range(len(ia) - 1):
        r[i] -= np.dot(a[ia[i]:ia[i + 1]], x[ja[ia[i]:ia[i + 1]]])
    assert "%.5e" % (np.linalg.norm(r) / np.linalg.norm(f)) == "8.82079e-07"


def _famg(lib_, ia, ja, a, f, p, x0=None):
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)) if x0 is None else x0.copy()
    bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x))
    fn = lib_.fasp_solver_famg
    fn.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.POINTER(T.AMG_param)]
    fn(C.byref(A), C.byref(bv), C.byref(xv), C.byref(p))
    return x


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("n", [12, 24])
def test_gpu_famg_solver_matches_reference(n):
    """fasp_solver_famg (SolFAMG.c:41): one full-multigrid cycle from a non-zero guess, against the compiled reference."""
    import faspsolver_amd as fa
    from _libs import ref
    ia, ja, a, f, ue = poisson7pt(n)
    x0 = np.cos(np.arange(len(f)) * 0.05) * 1e-3
    x1 = _famg(ref(), ia, ja, a, f, params(jac), x0)
    x2 = _famg(fa.lib(), ia, ja, a, f, params(jac), x0)
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()
    A, keep = T.as_csr(ia, ja, a)
    assert np.linalg.norm(f - _matvec(ia, ja, a, x2)) <= 0.2 * np.linalg.norm(f)   # one cycle: a solver step, not a solve


def _matvec(ia, ja, a, x):
    import scipy.sparse as sp
    return sp.csr_matrix((a, ja, ia), shape=(len(x), len(x))) @ x
