"""Edge cases of the drop-in entry points, each against the compiled reference (CPU: oracle vs reference)
and on the device (GPU vs oracle): trivial sizes, zero right-hand side, exact initial guess, iteration
limits, tolerances the solver cannot reach, unsymmetric input with the GMRES family, stop types."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import default_params, have_ref, orc_solve, poisson7pt, ref_solve

needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def _jac(itp, amgp):
    itp.tol = 1e-8; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667


def tridiag(n, lower=-1.0, diag=2.0, upper=-1.0):
    ia = [0]; ja = []; a = []
    for i in range(n):
        ja.append(i); a.append(diag)          # diagonal first, like the reference's generators
        if i > 0: ja.append(i - 1); a.append(lower)
        if i < n - 1: ja.append(i + 1); a.append(upper)
        ia.append(len(ja))
    return np.array(ia, dtype=np.int32), np.array(ja, dtype=np.int32), np.array(a)


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def one_by_one():
    ia, ja, a = tridiag(1); f = np.array([3.0])
    return ia, ja, a, f, None, _jac


@case
def two_by_two():
    ia, ja, a = tridiag(2); f = np.array([1.0, -2.0])
    return ia, ja, a, f, None, _jac


@case
def single_level_small():
    ia, ja, a = tridiag(40); f = np.linspace(-1, 1, 40)
    return ia, ja, a, f, None, _jac


@case
def zero_rhs():
    ia, ja, a, f, ue = poisson7pt(10)
    return ia, ja, a, np.zeros(len(f)), np.ones(len(f)), _jac


@case
def exact_initial_guess():
    """Integer data (P7(10): 726 and -121; x small integers): b = A x0 is exact in any summation order, the
    initial residual is exactly zero and the 'already converged' early exit returns 0 iterations."""
    ia, ja, a, f, ue = poisson7pt(10)
    x = np.random.default_rng(2).integers(-9, 10, len(f)).astype(np.float64)
    b = np.zeros(len(f))
    for i in range(len(f)):
        b[i] = np.dot(a[ia[i]:ia[i + 1]], x[ja[ia[i]:ia[i + 1]]])
    return ia, ja, a, b, x, _jac


@case
def guess_at_rounding_level_of_the_solution():
    """x0 = solution of a previous solve: the solver starts at the noise floor.  Whatever branch the safeguards
    take there depends on rounding, so only the outcome that matters is compared: x stays the solution."""
    ia, ja, a, f, ue = poisson7pt(10)
    i0, a0 = default_params(); _jac(i0, a0); i0.tol = 1e-13
    s0, x0, h0, rr0 = orc_solve(ia, ja, a, f, i0, a0)
    return ia, ja, a, f, x0, _jac


@case
def maxit_one():
    ia, ja, a, f, ue = poisson7pt(12)
    def mod(i, p): _jac(i, p); i.maxit = 1
    return ia, ja, a, f, None, mod


@case
def unreachable_tolerance():
    ia, ja, a, f, ue = poisson7pt(10)
    def mod(i, p): _jac(i, p); i.tol = 1e-30; i.maxit = 60
    return ia, ja, a, f, None, mod


@case
def unsymmetric_vgmres():
    ia, ja, a = tridiag(3000, lower=-1.3, diag=2.4, upper=-0.9)
    f = np.sin(np.arange(3000) * 0.01)
    def mod(i, p): _jac(i, p); i.itsolver_type = 5; i.restart = 20
    return ia, ja, a, f, None, mod


@case
def unsymmetric_vfgmres_precres():
    ia, ja, a = tridiag(3000, lower=-1.3, diag=2.4, upper=-0.9)
    f = np.cos(np.arange(3000) * 0.02)
    def mod(i, p): _jac(i, p); i.itsolver_type = 6; i.restart = 10; i.stop_type = 2
    return ia, ja, a, f, None, mod


@case
def bicgstab_poisson():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 2
    return ia, ja, a, f, None, mod


@case
def minres_poisson():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 3
    return ia, ja, a, np.cos(np.arange(len(f)) * 0.37), None, mod


@case
def minres_precres_guess():
    ia, ja, a, f, ue = poisson7pt(12)
    def mod(i, p): _jac(i, p); i.itsolver_type = 3; i.stop_type = 2
    return ia, ja, a, np.cos(np.arange(len(f)) * 0.37), np.sin(np.arange(len(f)) * 0.11), mod


@case
def gcg_poisson():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 7
    return ia, ja, a, np.cos(np.arange(len(f)) * 0.37), None, mod


@case
def gcg_maxit():
    ia, ja, a, f, ue = poisson7pt(12)
    def mod(i, p): _jac(i, p); i.itsolver_type = 7; i.maxit = 3
    return ia, ja, a, np.cos(np.arange(len(f)) * 0.37), None, mod


@case
def gcr_poisson_restart3():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 8; i.restart = 3
    return ia, ja, a, np.cos(np.arange(len(f)) * 0.37), None, mod


@case
def gcr_unsymmetric():
    ia, ja, a = tridiag(3000, lower=-1.3, diag=2.4, upper=-0.9)
    f = np.sin(np.arange(3000) * 0.01)
    def mod(i, p): _jac(i, p); i.itsolver_type = 8; i.restart = 20
    return ia, ja, a, f, None, mod


@case
def bicgstab_unsymmetric_tight():
    ia, ja, a = tridiag(3000, lower=-1.3, diag=2.4, upper=-0.9)
    f = np.sin(np.arange(3000) * 0.01)
    def mod(i, p): _jac(i, p); i.itsolver_type = 2; i.tol = 1e-13; i.maxit = 50
    return ia, ja, a, f, None, mod


@case
def bicgstab_maxit():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 2; i.maxit = 2
    return ia, ja, a, f, None, mod


@case
def gmres_fixed_restart_small():
    ia, ja, a, f, ue = poisson7pt(16)
    def mod(i, p): _jac(i, p); i.itsolver_type = 4; i.restart = 3
    return ia, ja, a, f, None, mod


@case
def gmres_fixed_unsymmetric_precres():
    ia, ja, a = tridiag(3000, lower=-1.3, diag=2.4, upper=-0.9)
    f = np.cos(np.arange(3000) * 0.02)
    def mod(i, p): _jac(i, p); i.itsolver_type = 4; i.restart = 6; i.stop_type = 2; i.tol = 1e-12; i.maxit = 80
    return ia, ja, a, f, None, mod


@case
def modrelres_stop():
    ia, ja, a, f, ue = poisson7pt(12)
    def mod(i, p): _jac(i, p); i.stop_type = 3
    return ia, ja, a, f, None, mod


@case
def two_levels_forced():
    ia, ja, a, f, ue = poisson7pt(12)
    def mod(i, p): _jac(i, p); p.max_levels = 2
    return ia, ja, a, f, None, mod


# Cases that drive the iteration to the rounding floor: WHICH safeguard stops it there (and after how many
# iterations) is decided by rounding noise, so only the result that matters -- the solution -- is compared.
NOISE_FLOOR = {"guess_at_rounding_level_of_the_solution", "unreachable_tolerance"}


def params(mod):
    itp, amgp = default_params(); mod(itp, amgp)
    return itp, amgp


@needs_ref
@pytest.mark.parametrize("name", list(CASES))
def test_oracle_equals_reference_on_edge_cases(name):
    ia, ja, a, f, x0, mod = CASES[name]()
    i1, a1 = params(mod); i2, a2 = params(mod)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1, x0)
    s2, x2, h2 = ref_solve(ia, ja, a, f, i2, a2, x0)
    if name in NOISE_FLOOR:
        assert np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()
        return
    assert s1 == s2, (s1, s2)
    assert np.array_equal(x1, x2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_equals_oracle_on_edge_cases(name):
    ia, ja, a, f, x0, mod = CASES[name]()
    i1, a1 = params(mod); i2, a2 = params(mod)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1, x0)
    x2 = np.zeros(len(f)) if x0 is None else x0.copy()
    s2 = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x2, i2, a2)
    if name not in NOISE_FLOOR:
        assert s2 == s1, (s1, s2)
    # error relative to the size of the iterates involved (zero rhs: the iterate decays from x0 to ~0)
    scale = max(np.abs(x1).max(), 0.0 if x0 is None else np.abs(x0).max(), 1e-300)
    assert np.abs(x1 - x2).max() <= 1e-9 * scale
