"""Matrix-free interface of the reference (fasp.h:1109 mxv_matfree, SolMatFree.c): the reference keeps OLDER
texts of CG and of the GMRES variants for it.  CPU: the oracle's restatements against the compiled reference,
bit for bit.  GPU: the device drivers against the oracle -- operators installed by fasp_solver_matfree_init
(resident in HBM), a host callback as the operator, the device AMG as the preconditioner."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import default_params, have_ref, oracle, poisson7pt, poisson7pt_bsr, ref
from test_plugin_krylov import diag_pc

needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
NAMES = {0: "fasp_solver_pcg", 1: "fasp_solver_pvgmres", 2: "fasp_solver_pvfgmres", 3: "fasp_solver_pbcgs",
         4: "fasp_solver_pgmres", 5: "fasp_solver_pminres", 6: "fasp_solver_pgcg"}
NO_RESTART = (0, 3, 5, 6)
MXV_FCT = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


class MF(C.Structure):
    _fields_ = [("data", C.c_void_p), ("fct", C.c_void_p)]


def rhs(n):
    return np.random.default_rng(5).standard_normal(n)


def orc_mf(which, ia, ja, a, f, fct=None, tol=1e-8, maxit=300, restart=30, stop=1):
    o = oracle()
    o.orc_krylov_mf.argtypes = [C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_void_p,
                                C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, T.c_double_p]
    A, keep = T.as_csr(ia, ja, a)
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x)); rr = C.c_double(0)
    st = o.orc_krylov_mf(which, C.byref(A), C.byref(bv), C.byref(xv), C.cast(fct, C.c_void_p) if fct else None, None,
                         tol, 1e-18, maxit, restart, stop, 0, C.byref(rr))
    return st, x, rr.value


def call_mf(lib_, which, mf, f, pc=None, tol=1e-8, maxit=300, restart=30, stop=1):
    fn = getattr(lib_, NAMES[which])
    base = [C.c_void_p, C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_void_p, C.c_double, C.c_double, C.c_int]
    fn.argtypes = base + ([C.c_short, C.c_short] if which in NO_RESTART else [C.c_short, C.c_short, C.c_short])
    fn.restype = C.c_int
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x))
    args = (C.addressof(mf), C.byref(bv), C.byref(xv), pc, tol, 1e-18, maxit)
    st = fn(*args, stop, 0) if which in NO_RESTART else fn(*args, restart, stop, 0)
    return st, x


@needs_ref
@pytest.mark.parametrize("with_pc", [False, True])
@pytest.mark.parametrize("stop", [1, 2, 3])
@pytest.mark.parametrize("which,restart", [(0, 30), (1, 30), (1, 5), (2, 30), (2, 4), (3, 30), (4, 30), (4, 6), (5, 30), (6, 30)])
def test_oracle_matfree_equals_reference(which, restart, stop, with_pc):
    ia, ja, a, f, ue = poisson7pt(10)
    f = rhs(len(f))
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_mf(which, ia, ja, a, f, fct, restart=restart, stop=stop)
    R = ref()
    A, keep = T.as_csr(ia, ja, a)
    mf = MF(); R.fasp_solver_matfree_init(1, C.byref(mf), C.byref(A))
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = call_mf(R, which, mf, f, C.cast(C.pointer(pc), C.c_void_p) if pc is not None else None, restart=restart, stop=stop)
    assert s1 == s2 and s1 > 5
    assert np.array_equal(x1, x2)


@pytest.mark.gpu
@pytest.mark.parametrize("with_pc", [False, True])
@pytest.mark.parametrize("which,restart,stop", [(0, 30, 1), (0, 30, 2), (0, 30, 3), (1, 30, 1), (1, 5, 1), (2, 30, 1), (2, 4, 2),
                                                (3, 30, 1), (4, 30, 1), (4, 6, 3), (5, 30, 1), (5, 30, 2), (5, 30, 3),
                                                (6, 30, 1)])
def test_gpu_matfree_csr_matches_oracle(which, restart, stop, with_pc):
    ia, ja, a, f, ue = poisson7pt(14)
    f = rhs(len(f))
    fct = diag_pc(ia, ja, a) if with_pc else None
    s1, x1, rr = orc_mf(which, ia, ja, a, f, fct, restart=restart, stop=stop)
    L = fa.lib()
    A, keep = T.as_csr(ia, ja, a)
    mf = MF(); L.fasp_solver_matfree_init(1, C.byref(mf), C.byref(A))
    pc = T.precond(None, fct) if with_pc else None
    s2, x2 = call_mf(L, which, mf, f, C.cast(C.pointer(pc), C.c_void_p) if pc is not None else None, restart=restart, stop=stop)
    if which == 3:
        # BiCGstab on a rough right-hand side is not a stable recurrence: the last bits of the dot products
        # (tree sums on the device, left-to-right sums in the reference) move the iteration count by a few
        # steps.  Both answers solve the system to the tolerance; they agree to cond(A) * tol.
        assert abs(s1 - s2) <= 4 and s2 > 5
        assert np.abs(x1 - x2).max() <= 1e-6 * np.abs(x1).max()
        return
    assert s1 == s2 and s1 > 5
    assert np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 3])
def test_gpu_matfree_host_callback_operator(which):
    """A caller-supplied y = A x (host function) as the operator: same iteration as the resident CSR."""
    ia, ja, a, f, ue = poisson7pt(10)
    f = rhs(len(f))
    n = len(f)
    s1, x1, rr = orc_mf(which, ia, ja, a, f)
    import scipy.sparse as sp
    M = sp.csr_matrix((a, ja, ia), shape=(n, n))

    def mxv(data, x, y):
        np.ctypeslib.as_array(y, (n,))[:] = M @ np.ctypeslib.as_array(x, (n,))
    cb = MXV_FCT(mxv)
    mf = MF(None, C.cast(cb, C.c_void_p))
    s2, x2 = call_mf(fa.lib(), which, mf, f)
    assert abs(s1 - s2) <= 1 and s2 > 5        # scipy's row sums may differ from the reference's in the last bit
    assert np.abs(x1 - x2).max() <= 1e-7 * np.abs(x1).max()


@pytest.mark.gpu
def test_gpu_matfree_bsr_and_dispatch():
    """fasp_solver_matfree_init(MAT_BSR) + fasp_solver_krylov (SolMatFree.c:157), as test/main/regression_mf.c does."""
    ia, ja, val, nb = poisson7pt_bsr(8)
    n = (len(ia) - 1) * nb
    f = rhs(n)
    L = fa.lib()
    A, keep = T.as_bsr(ia, ja, val, nb)
    mf = MF(); L.fasp_solver_matfree_init(2, C.byref(mf), C.byref(A))
    L.fasp_solver_krylov.argtypes = [C.c_void_p, C.POINTER(T.dvector), C.POINTER(T.dvector), C.POINTER(T.ITS_param)]
    out = {}
    for solver in (1, 2, 3, 4, 5, 6, 7):
        itp, _ = default_params(); itp.tol = 1e-8; itp.itsolver_type = solver; itp.restart = 30; itp.maxit = 500
        x = np.zeros(n); bv, fk = T.as_vec(f); xv = T.dvector(n, T.dp(x))
        st = L.fasp_solver_krylov(C.addressof(mf), C.byref(bv), C.byref(xv), C.byref(itp))
        assert st > 0, (solver, st)
        y = np.zeros(n)
        L.fasp_blas_dbsr_mxv(C.byref(A), T.dp(x), T.dp(y))
        assert np.linalg.norm(f - y) <= 1.2e-8 * np.linalg.norm(f), solver
        out[solver] = x


@pytest.mark.gpu
def test_gpu_matfree_pcg_with_device_amg():
    """fasp_solver_pcg(mf, ..., pc = device AMG): operator and preconditioner both resident."""
    ia, ja, a, f, ue = poisson7pt(20)
    itp, amgp = default_params(); itp.tol = 1e-8; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    x0 = np.zeros(len(f))
    s0 = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x0, itp, amgp)
    itp, amgp = default_params(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    L = fa.lib()
    A, keep = T.as_csr(ia, ja, a)
    pc = L.fasp_hip_precond_setup(C.byref(A), C.byref(amgp))
    mf = MF(); L.fasp_solver_matfree_init(1, C.byref(mf), C.byref(A))
    s1, x1 = call_mf(L, 0, mf, f, pc)
    L.fasp_hip_precond_free(pc)
    assert s1 == s0
    assert np.abs(x1 - x0).max() <= 1e-10 * np.abs(x0).max()
