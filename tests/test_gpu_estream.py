"""k_csr_estream (csrc/kernels3.hip.h, round 6): the entry-parallel stream kernel of the long-row operators, through the C-ABI
(fasp_blas_dcsr_mxv / _aAxpy, BlaSpmvCSR.c:242 / :494; fasp_smoother_dcsr_jacobi, ItrSmootherCSR.c:98) against the CPU oracle's row
sums on the same inputs, and against the row kernel it replaces (fasp_hip_tune("estream", 0)).

The kernel cuts the ENTRIES of an operator into equal wave ranges, whatever the rows do -- so the cases are the row shapes that make
that hard: rows longer than several wave ranges (parts summed by the last of several waves), empty rows at the start, at the end and
at chunk boundaries, runs of hundreds of one- and two-entry rows inside one chunk (beyond the staged row-pointer window), operators
with more than 65536 columns (16-bit columns relative to a per-row base), ranges that end exactly on a row boundary.

Tolerance: a row sum in any fixed order agrees with the reference's left-to-right sum to 1e-13 of the row's absolute sum.
Determinism: two launches give the same bits (the parts of a cut row are summed in wave order by whichever wave arrives last)."""
import ctypes as C

import numpy as np
import pytest

from _libs import T, oracle

pytestmark = pytest.mark.gpu


def _matrix(lens, m, seed, band=None):
    """Random CSR with the given row lengths; columns drawn without repetition from [0, m) (band: from a window of that width around
    the row's own position, for operators with more than 65536 columns whose rows span less)."""
    rng = np.random.default_rng(seed)
    n = len(lens)
    ia = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = []
    for i, l in enumerate(lens):
        if l == 0:
            continue
        if band is None:
            c = rng.choice(m, size=l, replace=False)
        else:
            lo = int(min(max(0, i * (m / n) - band / 2), m - band))
            c = lo + rng.choice(band, size=l, replace=False)
        if i < m and l > 1:
            c[0] = i if i not in c else c[0]     # a diagonal where there is room for one (Jacobi)
        rng.shuffle(c)
        cols.append(c)
    ja = np.concatenate(cols).astype(np.int32)
    a = rng.standard_normal(len(ja))
    return ia, ja, a


def _cases():
    rng = np.random.default_rng(42)
    out = {}
    # (a) plain long rows, <= 65536 columns
    out["long-rows"] = (rng.integers(50, 1200, 700), 30000, None)
    # (b) a few rows longer than several wave ranges (nnz ~ 300 K, 32 ranges of ~9 K entries), empty rows around them and at both ends
    lens = rng.integers(60, 400, 900)
    lens[[0, 1, 2, 450, 451, 898, 899]] = 0
    lens[[100, 452, 700]] = [40000, 25000, 52000]
    out["rows-across-waves"] = (lens, 60000, None)
    # (c) runs of one- and two-entry rows (more rows in a chunk than the staged pointer window holds) between long rows
    lens = rng.integers(100, 900, 600)
    lens[200:520] = rng.integers(0, 3, 320)
    out["short-row-runs"] = (lens, 20000, None)
    # (d) more than 65536 columns, every row inside a band of 40000: 16-bit columns relative to the row's smallest
    out["relative-columns"] = (rng.integers(50, 700, 1500), 200000, 40000)
    # (e) equal rows of 512 entries: ranges and chunks end exactly on row boundaries
    out["aligned"] = (np.full(640, 512), 50000, None)
    # (f), (g) the row lengths of the levels the kernel is SELECTED for (mean below 256: four and eight lanes per row), entry counts that
    # are no multiple of 8 (the last 16-byte piece of the arrays reaches into their slack)
    out["rows-of-100"] = (rng.integers(48, 160, 1501), 9000, None)
    out["rows-of-180"] = (rng.integers(130, 250, 971), 971, None)
    return out


CASES = _cases()


@pytest.mark.parametrize("name", list(CASES))
def test_estream_matches_oracle_and_row_kernel(gpu, name):
    lens, m, band = CASES[name]
    ia, ja, a = _matrix(lens, m, seed=len(name), band=band)
    n = len(lens)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(m)
    A, keep = T.as_csr(ia, ja, a, ncol=m)
    L = gpu.lib()
    assert len(a) >= 65536 and len(a) / n > 48      # the operator class the kernel serves
    rowabs = np.array([np.sum(np.abs(a[ia[i]:ia[i + 1]] * x[ja[ia[i]:ia[i + 1]]])) for i in range(n)])
    y_ref = np.zeros(n)
    oracle().orc_mxv(C.byref(A), T.dp(x), T.dp(y_ref))
    got = {}
    try:
        for es in (2, 0):   # 2: the entry stream wherever its tables exist, 0: the row kernel
            L.fasp_hip_tune(b"estream", es)
            # (poison what freed device memory holds: an uninitialised slack behind the operator's arrays must not matter)
            junk = np.full(1 << 20, -0.5)
            L.fasp_blas_darray_ax.argtypes = [C.c_int, C.c_double, T.c_double_p]
            L.fasp_blas_darray_ax(len(junk), 2.0, T.dp(junk))
            y = np.full(n, 7.0)
            L.fasp_blas_dcsr_mxv(C.byref(A), T.dp(x), T.dp(y))
            assert np.all(y[lens == 0] == 0.0)
            assert np.all(np.abs(y - y_ref) <= 1e-13 * np.maximum(rowabs, 1e-300) + 1e-300), (name, es)
            y2 = np.full(n, -3.0)
            L.fasp_blas_dcsr_mxv(C.byref(A), T.dp(x), T.dp(y2))
            assert np.array_equal(y, y2), (name, es)                      # deterministic, launch to launch
            got[es] = y
            for alpha in (1.0, -1.0, 0.7):
                y0 = rng.standard_normal(n)
                y1 = y0.copy(); y3 = y0.copy()
                oracle().orc_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y1))
                L.fasp_blas_dcsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y3))
                assert np.all(np.abs(y1 - y3) <= 1e-13 * (rowabs + np.abs(y0)) + 1e-300), (name, es, alpha)
    finally:
        L.fasp_hip_tune(b"estream", 1)
    assert np.all(np.abs(got[0] - got[2]) <= 2e-13 * np.maximum(rowabs, 1e-300) + 1e-300)


@pytest.mark.parametrize("name", ["long-rows", "rows-across-waves", "relative-columns", "rows-of-100", "rows-of-180"])
def test_estream_jacobi_sweeps(gpu, name):
    """Weighted Jacobi (the diagonal entry is left out of the row sum by comparing columns: the device copy is sorted by column) on a
    square, diagonally dominant variant of the operator."""
    lens, m, band = CASES[name]
    n = len(lens)
    lens = np.minimum(lens, n - 1)
    ia, ja, a = _matrix(lens, n, seed=7 + len(name), band=min(band, n) if band else None)
    for i in range(n):       # make every stored diagonal dominant
        kb, ke = ia[i], ia[i + 1]
        hit = np.nonzero(ja[kb:ke] == i)[0]
        if len(hit):
            a[kb + hit[0]] = 1.0 + np.sum(np.abs(a[kb:ke]))
    A, keep = T.as_csr(ia, ja, a, ncol=n)
    rng = np.random.default_rng(11)
    f = rng.standard_normal(n)
    L = gpu.lib()
    try:
        for es in (2, 0):
            L.fasp_hip_tune(b"estream", es)
            u1 = rng.standard_normal(n); u2 = u1.copy()
            oracle().orc_smoother_jacobi(T.dp(u1), 0, n - 1, 1, C.byref(A), T.dp(f), 3, 0.8)
            uv = T.dvector(n, T.dp(u2)); bv = T.dvector(n, T.dp(f))
            L.fasp_smoother_dcsr_jacobi(C.byref(uv), n - 1, 0, -1, C.byref(A), C.byref(bv), 3, 0.8)
            assert np.allclose(u1, u2, rtol=1e-12, atol=1e-13 * np.max(np.abs(u1))), (name, es)
    finally:
        L.fasp_hip_tune(b"estream", 1)
