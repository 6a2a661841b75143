"""The safe CG of coarsest levels of at most 128 rows (KrySPcg.c:60-365 as fasp_coarse_itsolver calls it, AuxParam/SolWrapper: tol =
coarse tolerance, MaxIt = max(250, min(n^2, 1000))): the three single-workgroup forms of csrc/small_solvers.hip.h -- k_spcg_wave (the
matrix dense in LDS), k_spcg_reg (in registers, p broadcast from LDS), k_spcg_dpp (in registers as 16 x 16 blocks, p broadcast inside
the multiply-adds; form 4, the default: the same with the next direction sent before the tests of the iteration) -- inside whole solves against the CPU oracle.

Config 5 of BASELINE.json (SA-AMG, W-cycle, VFGMRES(30) on the anisotropic 27-point operator) at sizes whose coarsest levels have 44, 80,
126 and 128 rows: the three instantiations of k_spcg_dpp (<4,4>, <6,2>, <8,2>), one of them with every row in use.  A W-cycle visits
the coarsest level twice per cycle, the second time with a nonzero iterate: both entries of the kernels (x_zero and the initial
residual product) are on the path.

Bar: outer iteration count and residual history as the oracle's (the coarse solve is an inner iteration run to its own tolerance: sums
in another order move its result by O(1e-16) relative and its iteration counts by a few in a thousand), the solution to 1e-9."""
import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T
import _libs

pytestmark = pytest.mark.gpu


def _params(amg_tol=None):
    itp, amgp = _libs.default_params()
    itp.tol = 1e-8; itp.itsolver_type = 6; itp.restart = 30
    amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE
    amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    if amg_tol is not None:
        amgp.tol = amg_tol
    return itp, amgp


@pytest.mark.parametrize("n,rows", [(48, 44), (10, 80), (12, 126), (28, 128)])
def test_small_coarse_cg_forms_match_oracle(gpu, n, rows):
    ia, ja, a, f = fa.aniso27pt(n)
    itp, amgp = _params()
    s_ref, x_ref, h_ref, rr_ref = _libs.orc_solve(ia, ja, a, f, *_params())
    L = gpu.lib()
    got = {}
    try:
        for form in (4, 3, 2, 1):
            L.fasp_hip_tune(b"small_onewave", form)
            H = fa.AMG(ia, ja, a, amgp)
            assert H.matrix(H.num_levels - 1, 0)[0] == rows
            st, x, hist, stats = H.solve(f, itp)
            H.close()
            assert st == s_ref, (form, st, s_ref)
            assert abs(stats.relres - rr_ref) <= 1e-6 * rr_ref, (form, stats.relres, rr_ref)
            assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max(), form
            got[form] = stats.coarse_iters
    finally:
        L.fasp_hip_tune(b"small_onewave", 4)
    assert got[4] > 0
    for form in (3, 2, 1):
        assert abs(got[form] - got[4]) <= 0.02 * got[4] + 2, got


@pytest.mark.parametrize("n", [10, 48])
def test_small_coarse_cg_with_a_tolerance_it_cannot_reach(gpu, n):
    """AMG_param.tol = 1e-14 makes the coarse tolerance 1e-18 (PreMGCycle.c: tol * 1e-4): no coarse solve converges by its residual
    test, every one of them ends through the reference's stagnation and safety branches (Checks I-III with their true-residual
    products and restarts, KrySPcg.c:195-300) -- in k_spcg_dpp's send-ahead form the branches in which a product sent ahead is
    dropped and the iteration count and (z, r) step back.  Same verdicts as the oracle in all forms; forms 3 and 4 (the same
    arithmetic with and without sending ahead) agree to the last bit."""
    ia, ja, a, f = fa.aniso27pt(n)
    s_ref, x_ref, h_ref, rr_ref = _libs.orc_solve(ia, ja, a, f, *_params(1e-14))
    L = gpu.lib()
    got, sol = {}, {}
    try:
        for form in (4, 3, 2, 1):
            L.fasp_hip_tune(b"small_onewave", form)
            itp, amgp = _params(1e-14)
            H = fa.AMG(ia, ja, a, amgp)
            st, x, hist, stats = H.solve(f, itp)
            H.close()
            assert st == s_ref, (form, st, s_ref)
            assert abs(stats.relres - rr_ref) <= 1e-6 * rr_ref, (form, stats.relres, rr_ref)
            assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max(), form
            got[form] = stats.coarse_iters; sol[form] = x
    finally:
        L.fasp_hip_tune(b"small_onewave", 4)
    assert got[4] == got[3] and np.array_equal(sol[4], sol[3])
    for form in (2, 1):
        assert abs(got[form] - got[4]) <= 0.02 * got[4] + 2, got
