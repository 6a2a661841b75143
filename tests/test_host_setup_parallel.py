"""The host setup's parallel loops must reproduce the serial text exactly: the stable counting transposes
(BlaSparseCSR.c:875 / :952) run as a two-level radix pass above a size threshold; here the threshold is
forced to 0 and every operator of the hierarchy is compared bit for bit with the serial build."""
import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T


def _build(ia, ja, a, kind, minnz):
    fa.lib().fasp_hip_tune(b"host_parallel_min", minnz)
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    amgp.AMG_type = {"rs": T.CLASSIC_AMG, "ac": T.CLASSIC_AMG, "sa": T.SA_AMG, "ua": T.UA_AMG}[kind]
    if kind == "ac":   # aggressive coarsening, two-path couplings between the C points
        amgp.coarsening_type = T.COARSE_AC; amgp.aggressive_path = 2
    return fa.AMG(ia, ja, a, amgp, host_only=True)


@pytest.mark.parametrize("kind,n", [("rs", 24), ("rs", 33), ("ac", 33), ("sa", 24), ("ua", 20)])
def test_parallel_transposes_reproduce_the_serial_hierarchy(kind, n):
    ia, ja, a, f, ue = fa.poisson7pt(n)
    try:
        H1 = _build(ia, ja, a, kind, 2**31 - 1)
        H2 = _build(ia, ja, a, kind, 0)
    finally:
        fa.lib().fasp_hip_tune(b"host_parallel_min", 1 << 20)
    assert H1.num_levels == H2.num_levels and H1.num_levels >= 2
    for l in range(H1.num_levels):
        for w in (0, 1, 2):
            if w and l == H1.num_levels - 1:
                continue
            m1, m2 = H1.matrix(l, w), H2.matrix(l, w)
            assert m1[0] == m2[0] and m1[1] == m2[1]
            assert all(np.array_equal(x, y) for x, y in zip(m1[2:], m2[2:])), (l, w)
    H1.close(); H2.close()
