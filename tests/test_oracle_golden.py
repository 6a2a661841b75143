"""The oracle against the reference's OWN golden logs and against the reference's
numbers recorded in BASELINE.md / SURVEY.md section 6 (oracle probes of the serial build).

Golden sources (paths relative to the reference tree):
  test/out/reg.out:252-257    AMG-PCG on csrmat_FD  -> 1 iteration,  relres 4.938174e-15
  test/out/reg.out:574-579    AMG-PCG on csrmat_FE  -> 6 iterations, relres 2.728796e-11
  tutorial/out/poisson-pcg-c.out  csrmat_FE, defaults, tol 1e-6: levels 3969/1985/541/141
                              (27281/28523/7951/1803 nnz), 4 iterations, residual history
  test/main/regression.c:24-36    check_solu: max-diff to the shipped solution < 1e-4
"""
import ctypes as C

import numpy as np
import pytest

from _libs import (DATA, OrcAMG, T, default_params, oracle, orc_solve, poisson7pt, read_csr,
                   read_vec, read_vecind)


def _fe():
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat")
    return ia, ja, a, read_vec(DATA + "/rhs_FE.dat"), read_vecind(DATA + "/sol_FE.dat")


def _fd():
    ia, ja, a = read_csr(DATA + "/csrmat_FD.dat")
    return ia, ja, a, read_vec(DATA + "/rhs_FD.dat"), read_vecind(DATA + "/sol_FD.dat")


def test_regression_fd_golden():
    ia, ja, a, f, sol = _fd()
    itp, amgp = default_params()
    itp.tol = 1e-10; itp.maxit = 500  # regression.c:61-62
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == 1
    assert f"{rr:.6e}" == "4.938174e-15"           # reg.out:255
    assert np.max(np.abs(x - sol)) < 1e-4           # check_solu


def test_regression_fe_golden():
    ia, ja, a, f, sol = _fe()
    itp, amgp = default_params()
    itp.tol = 1e-10; itp.maxit = 500
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == 6
    assert f"{rr:.6e}" == "2.728796e-11"           # reg.out:577
    assert f"{rr:.10e}" == "2.7287964242e-11"      # SURVEY.md section 4 (oracle probe)
    assert np.max(np.abs(x - sol)) < 1e-4


def test_tutorial_poisson_pcg_golden():
    ia, ja, a, f, sol = _fe()
    itp, amgp = default_params()  # tol 1e-6, GS + C/F order, V(1,1): the tutorial log's parameters
    A, keep = T.as_csr(ia, ja, a)
    H = OrcAMG(A, amgp)
    rows = [H.level(l).A.row for l in range(H.num_levels)]
    nnzs = [H.level(l).A.nnz for l in range(H.num_levels)]
    assert rows == [3969, 1985, 541, 141]
    assert nnzs == [27281, 28523, 7951, 1803]
    H.free()
    itp, amgp = default_params()
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == 4
    rel = hist[:5] / hist[0]
    assert [f"{v:.6e}" for v in rel[1:]] == ["1.156153e-02", "3.127181e-04", "4.813471e-06", "5.312526e-08"]
    assert [f"{v:.6e}" for v in hist[:5]] == ["7.514358e+00", "8.687750e-02", "2.349876e-03",
                                              "3.617014e-05", "3.992022e-07"]


# reference (serial libfasp) on P7(n), PCG tol 1e-8, classical AMG
# defaults + SMOOTHER_JACOBI.  (n, relaxation, levels, iterations, final relres %.10e)
BASELINE_ROWS = [(16, 0.6667, 3, 8, "1.2526544014e-09"),
                 (32, 1.0, 5, 9, "3.3131623388e-09"),
                 (48, 0.6667, 6, 9, "2.2240037406e-09"),
                 (64, 0.6667, 6, 9, "3.0782769324e-09")]


@pytest.mark.parametrize("n,w,levels,its,relres", BASELINE_ROWS)
def test_baseline_table(n, w, levels, its, relres):
    ia, ja, a, f, ue = poisson7pt(n)
    itp, amgp = default_params()
    itp.tol = 1e-8; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = w
    A, keep = T.as_csr(ia, ja, a)
    p2 = T.AMG_param.from_buffer_copy(amgp)
    H = OrcAMG(A, p2)
    assert H.num_levels == levels
    H.free()
    st, x, hist, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == its
    assert f"{rr:.10e}" == relres


def test_generator_counts():
    # nnz = 7 n^3 - 6 n^2; diagonal first in every row
    for n in (3, 5, 8):
        ia, ja, a, f, ue = poisson7pt(n)
        assert len(a) == 7 * n ** 3 - 6 * n ** 2
        assert np.all(ja[ia[:-1]] == np.arange(n ** 3))
        h = 1.0 / (n + 1)
        assert np.all(a[ia[:-1]] == 2.0 * (1.0 / (h * h) * 3))
