"""Plug-in level of the reference: fasp_solver_dcsr_pcg / _pvgmres / _pvfgmres with a
caller-supplied `precond` (fasp.h:1095), and the device AMG preconditioner handed out as a `precond`."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import default_params, have_ref, oracle, orc_solve, poisson7pt, ref

needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
NO_RESTART = (0, 3, 5, 6)   # which: 0 PCG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES, 5 MinRes, 6 GCG, 7 GCR
KARGS = [C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_void_p, C.c_double, C.c_double,
         C.c_int]


def diag_pc(ia, ja, a):
    """z = D^-1 r as a host callback (the shape of a user preconditioner)."""
    n = len(ia) - 1
    d = np.array([a[ia[i]:ia[i + 1]][ja[ia[i]:ia[i + 1]] == i][0] for i in range(n)])

    def fct(r, z, data):
        rv = np.ctypeslib.as_array(r, (n,)); zv = np.ctypeslib.as_array(z, (n,))
        zv[:] = rv / d
    return T.PRECOND_FCT(fct)


def orc_krylov(which, ia, ja, a, f, fct=None, tol=1e-8, maxit=500, restart=30, stop=1):
    o = oracle()
    o.orc_krylov_dcsr.argtypes = [C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector),
                                  C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                  C.c_int, T.c_double_p]
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x)); rr = C.c_double(0)
    st = o.orc_krylov_dcsr(which, C.byref(A), C.byref(bv), C.byref(xv), C.cast(fct, C.c_void_p) if fct else None,
                           None, tol, 1e-18, maxit, restart, stop, 0, C.byref(rr))
    return st, x, rr.value


def ref_krylov(which, ia, ja, a, f, pc=None, tol=1e-8, maxit=500, restart=30, stop=1):
    R = ref()
    fn = [R.fasp_solver_dcsr_pcg, R.fasp_solver_dcsr_pvgmres, R.fasp_solver_dcsr_pvfgmres, R.fasp_solver_dcsr_pbcgs, R.fasp_solver_dcsr_pgmres,
          R.fasp_solver_dcsr_pminres, R.fasp_solver_dcsr_pgcg, R.fasp_solver_dcsr_pgcr][which]
    fn.argtypes = KARGS + ([C.c_short, C.c_short] if which in NO_RESTART else [C.c_short, C.c_short, C.c_short])
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x))
    args = (C.byref(A), C.byref(bv), C.byref(xv), C.cast(pc, C.c_void_p) if pc is not None else None, tol, 1e-18, maxit)
    st = fn(*args, stop, 0) if which in NO_RESTART else fn(*args, restart, stop, 0)
    return st, x


def gpu_krylov(which, ia, ja, a, f, pc=None, tol=1e-8, maxit=500, restart=30, stop=1):
    L = fa.lib()
    fn = [L.fasp_solver_dcsr_pcg, L.fasp_solver_dcsr_pvgmres, L.fasp_solver_dcsr_pvfgmres, L.fasp_solver_dcsr_pbcgs, L.fasp_solver_dcsr_pgmres,
          L.fasp_solver_dcsr_pminres, L.fasp_solver_dcsr_pgcg, L.fasp_solver_dcsr_pgcr][which]
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x))
    args = (C.byref(A), C.byref(bv), C.byref(xv), pc, tol, 1e-18, maxit)
    st = fn(*args, stop, 0) if which in NO_RESTART else fn(*args, restart, stop, 0)
    return st, x


@needs_ref
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("with_pc", [False, True])
def test_oracle_plugin_krylov_equals_reference(which, with_pc):
    ia, ja, a, f, ue = poisson7pt(10)
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_krylov(which, ia, ja, a, f, fct)
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = ref_krylov(which, ia, ja, a, f, C.pointer(pc) if pc is not None else None)
    assert s1 == s2 and s1 > 0
    assert np.array_equal(x1, x2)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("with_pc", [False, True])
def test_gpu_plugin_krylov_matches_oracle(which, with_pc):
    ia, ja, a, f, ue = poisson7pt(16)
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_krylov(which, ia, ja, a, f, fct)
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = gpu_krylov(which, ia, ja, a, f, C.byref(pc) if pc is not None else None)
    assert s1 == s2
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()


def _jac(itp, amgp):
    itp.tol = 1e-8; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667


def _rand_rhs(n):
    return np.random.default_rng(7).standard_normal(n)


@needs_ref
@pytest.mark.parametrize("which,restart", [(5, 30), (6, 30), (7, 30), (7, 5)])
@pytest.mark.parametrize("with_pc", [False, True])
@pytest.mark.parametrize("stop", [1, 2, 3])
def test_oracle_minres_gcg_gcr_equal_reference(which, restart, with_pc, stop):
    """KryPminres.c:61, KryPgcg.c:60, KryPgcr.c:55 on a right-hand side that is not an eigenvector
    (the generator's f converges in one step), restart cycles of GCR included: bit for bit."""
    ia, ja, a, f, ue = poisson7pt(10)
    f = _rand_rhs(len(f))
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_krylov(which, ia, ja, a, f, fct, maxit=200, restart=restart, stop=stop)
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = ref_krylov(which, ia, ja, a, f, C.pointer(pc) if pc is not None else None, maxit=200, restart=restart, stop=stop)
    assert s1 == s2 and s1 > 5
    assert np.array_equal(x1, x2)


@pytest.mark.gpu
@pytest.mark.parametrize("which,restart", [(5, 30), (6, 30), (7, 30), (7, 5)])
@pytest.mark.parametrize("with_pc", [False, True])
def test_gpu_minres_gcg_gcr_match_oracle(which, restart, with_pc):
    ia, ja, a, f, ue = poisson7pt(14)
    f = _rand_rhs(len(f))
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_krylov(which, ia, ja, a, f, fct, maxit=300, restart=restart)
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = gpu_krylov(which, ia, ja, a, f, C.byref(pc) if pc is not None else None, maxit=300, restart=restart)
    assert s1 == s2 and s1 > 5
    assert np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [3, 7, 8])
def test_gpu_dropin_dispatches_minres_gcg_gcr(solver):
    """fasp_solver_dcsr_krylov_amg with itsolver_type MinRes / GCG / GCR (SolCSR.c:98/:118/:123)."""
    ia, ja, a, f, ue = poisson7pt(20)
    f = _rand_rhs(len(f))
    itp, amgp = default_params(); _jac(itp, amgp); itp.itsolver_type = solver
    s0, x0, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    itp, amgp = default_params(); _jac(itp, amgp); itp.itsolver_type = solver
    x1 = np.zeros(len(f))
    s1 = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x1, itp, amgp)
    assert s1 == s0 and s1 > 0
    assert np.abs(x1 - x0).max() <= 1e-9 * np.abs(x0).max()


@pytest.mark.gpu
@pytest.mark.parametrize("which,solver", [(0, 1), (1, 5), (2, 6), (3, 2), (4, 4), (5, 3), (6, 7), (7, 8)])
def test_gpu_amg_as_precond_equals_dropin(which, solver):
    """fasp_precond_setup + fasp_solver_dcsr_pcg (tutorial/main/poisson-pcg.c:81,91) is the same
    computation as fasp_solver_dcsr_krylov_amg."""
    ia, ja, a, f, ue = poisson7pt(20)
    itp, amgp = default_params(); _jac(itp, amgp); itp.itsolver_type = solver
    x1 = np.zeros(len(f))
    s1 = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x1, itp, amgp)
    itp, amgp = default_params(); _jac(itp, amgp)
    A, keep = T.as_csr(ia, ja, a)
    pc = fa.lib().fasp_hip_precond_setup(C.byref(A), C.byref(amgp))
    assert pc
    s2, x2 = gpu_krylov(which, ia, ja, a, f, pc)
    fa.lib().fasp_hip_precond_free(pc)
    assert s1 == s2 and s1 > 0
    # same computation up to the row-sum order of the deep levels: the resident handle keeps their rows
    # sorted by column, the one-shot drop-in call skips that re-sorting
    assert np.abs(x1 - x2).max() <= 1e-12 * np.abs(x1).max()


@pytest.mark.gpu
@needs_ref
def test_gpu_precond_plugged_into_reference_cpu_pcg():
    """The device AMG as a `precond` inside the REFERENCE's own CPU PCG (it only calls pc->fct)."""
    ia, ja, a, f, ue = poisson7pt(16)
    itp, amgp = default_params(); _jac(itp, amgp)
    s0, x0, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    itp, amgp = default_params(); _jac(itp, amgp)
    A, keep = T.as_csr(ia, ja, a)
    pc = fa.lib().fasp_hip_precond_setup(C.byref(A), C.byref(amgp))
    s1, x1 = ref_krylov(0, ia, ja, a, f, pc)
    fa.lib().fasp_hip_precond_free(pc)
    assert s1 == s0
    assert np.abs(x1 - x0).max() <= 1e-10 * np.abs(x0).max()
