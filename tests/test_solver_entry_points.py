"""The solver-level entry points next to the AMG drop-in: fasp_solver_dcsr_itsolver / _krylov / _krylov_diag
(SolCSR.c:56/:245/:333) and their block twins (SolBSR.c:64/:145/:186) -- what test/main/regression.c drives for its
"CG solver", "Diagonal preconditioned CG solver", "... in BSR format" problems.  Checked on the GPU against the compiled
reference (same entry points, CPU) where it is available, else against the oracle's Krylov restatements."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import default_params, have_ref, poisson7pt, poisson7pt_bsr, ref
from test_plugin_krylov import diag_pc, orc_krylov

pytestmark = pytest.mark.gpu
WHICH = {1: 0, 2: 3, 3: 5, 4: 4, 5: 1, 6: 2, 7: 6, 8: 7}   # itsolver_type -> oracle `which`


def _rhs(n):
    return np.random.default_rng(11).standard_normal(n)


def _call(lib_, name, A, f, itp):
    fn = getattr(lib_, name)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(T.dvector), C.POINTER(T.dvector), C.POINTER(T.ITS_param)]
    x = np.zeros(len(f)); bv, fk = T.as_vec(f); xv = T.dvector(len(f), T.dp(x))
    st = fn(C.addressof(A), C.byref(bv), C.byref(xv), C.byref(itp))
    return st, x


@pytest.mark.parametrize("solver", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("entry", ["fasp_solver_dcsr_krylov", "fasp_solver_dcsr_krylov_diag"])
def test_csr_krylov_entry_points(entry, solver):
    ia, ja, a, f, ue = poisson7pt(12)
    f = _rhs(len(f))
    itp, _ = default_params(); itp.tol = 1e-8; itp.itsolver_type = solver; itp.restart = 20; itp.maxit = 400
    A, keep = T.as_csr(ia, ja, a)
    s2, x2 = _call(fa.lib(), entry, A, f, itp)
    fct = diag_pc(ia, ja, a) if entry.endswith("diag") else None
    s1, x1, rr = orc_krylov(WHICH[solver], ia, ja, a, f, fct, maxit=400, restart=20)
    if solver == 2:   # BiCGstab on a rough right-hand side: the iteration count moves with the last bits of the dots
        assert abs(s1 - s2) <= 4 and s2 > 0
        assert np.abs(x1 - x2).max() <= 1e-6 * np.abs(x1).max()
    else:
        assert s1 == s2 and s1 > 5
        assert np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()
    if have_ref():
        itp2, _ = default_params(); itp2.tol = 1e-8; itp2.itsolver_type = solver; itp2.restart = 20; itp2.maxit = 400
        s3, x3 = _call(ref(), entry, A, f, itp2)
        assert s3 == s1 and np.array_equal(x3, x1)   # the oracle's restatement IS the reference's entry point


@pytest.mark.parametrize("solver", [1, 4, 5, 6])
@pytest.mark.parametrize("entry", ["fasp_solver_dbsr_krylov", "fasp_solver_dbsr_krylov_diag"])
def test_bsr_krylov_entry_points(entry, solver):
    if not have_ref():
        pytest.skip("block Krylov entry points are checked against the compiled reference")
    ia, ja, val, nb = poisson7pt_bsr(8)
    n = (len(ia) - 1) * nb
    f = _rhs(n)
    A, keep = T.as_bsr(ia, ja, val, nb)
    itp, _ = default_params(); itp.tol = 1e-8; itp.itsolver_type = solver; itp.restart = 20; itp.maxit = 600
    itp2, _ = default_params(); itp2.tol = 1e-8; itp2.itsolver_type = solver; itp2.restart = 20; itp2.maxit = 600
    s1, x1 = _call(ref(), entry, A, f, itp2)
    s2, x2 = _call(fa.lib(), entry, A, f, itp)
    assert s1 == s2 and s1 > 5
    assert np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()


def test_unknown_solver_type_is_refused():
    ia, ja, a, f, ue = poisson7pt(6)
    A, keep = T.as_csr(ia, ja, a)
    itp, _ = default_params(); itp.itsolver_type = 13   # SOLVER_SMinRes belongs to fasp_solver_dcsr_itsolver_s
    st, x = _call(fa.lib(), "fasp_solver_dcsr_krylov", A, f, itp)
    assert st == T.ERROR_SOLVER_TYPE and not x.any()
