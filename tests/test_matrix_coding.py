"""Host side of the lossless matrix coding: the coder + decoder round trip is exact,
and the right operators qualify.  No GPU needed (fasp_hip_coding_selftest is pure host code)."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA, poisson7pt, read_csr


def coding(ia, ja, a, ncol=None):
    A, keep = T.as_csr(ia, ja, a)
    if ncol is not None:
        A.col = ncol
    k = C.c_int(-1)
    st = fa.lib().fasp_hip_coding_selftest(C.byref(A), C.byref(k))
    return st, k.value


def test_stencil_levels_are_pattern_coded_and_exact():
    ia, ja, a, f, ue = poisson7pt(24)
    assert coding(ia, ja, a) == (0, 5)          # 27 distinct rows
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI
    H = fa.AMG(ia, ja, a, amgp, host_only=True)
    kinds = {}
    for l in range(H.num_levels):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == H.num_levels - 1:
                continue
            r, c, i2, j2, v = H.matrix(l, which)
            st, k = coding(i2, j2, v, c)
            assert st == 0, (l, nm)             # whatever the coding, the round trip is exact
            kinds[(l, nm)] = k
    assert kinds[(0, "A")] == 5 and kinds[(0, "P")] == 5 and kinds[(0, "R")] == 5
    assert kinds[(1, "A")] == 5
    H.close()


def test_unstructured_matrix_stays_plain_csr():
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat")  # 3969 rows of an unstructured FE mesh
    st, k = coding(ia, ja, a)
    assert st == 0 and k == 0


def test_byte_dictionary_when_rows_do_not_repeat():
    """Few (offset, value) pairs but every row different: band matrix with a pseudo-random pattern."""
    n = 6000
    rng = np.random.default_rng(5)
    rows = []
    offs = np.array([-7, -3, -1, 0, 1, 2, 5, 11])
    vals = np.array([1.0, -2.0, 0.5])
    ia = [0]; ja = []; a = []
    for r in range(n):
        pick = np.sort(rng.choice(len(offs), size=rng.integers(3, 8), replace=False))
        for o in offs[pick]:
            c = r + o
            if 0 <= c < n:
                ja.append(c); a.append(vals[rng.integers(0, 3)])
        ia.append(len(ja))
    st, k = coding(np.array(ia, dtype=np.int32), np.array(ja, dtype=np.int32), np.array(a))
    assert st == 0 and k == 4


def test_signed_zero_and_nan_payloads_survive():
    ia, ja, a, f, ue = poisson7pt(18)
    a = a.copy()
    a[5] = -0.0
    a[77] = np.float64.fromhex("0x1.0000000000001p+0")
    st, k = coding(ia, ja, a)
    assert st == 0 and k in (4, 5)
