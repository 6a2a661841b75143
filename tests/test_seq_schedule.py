"""Host logic of the sequential sweeps' split form (csrc/seq_sched.cpp; no GPU): fasp_hip_seq_schedule_selftest builds the
schedule of a sweep -- strips, chunks, slots with LDS indices, ghost lists, tails, the rest CSR -- and walks it on the host the way
k_tri_flow / k_tri_level do (strips in ticket order, every operand through its LDS index, operands must come from earlier chunks
or earlier strips), then compares with the plain sequential Gauss-Seidel sweep of ItrSmootherCSR.c:251 over the same rows."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T


def _selftest(ia, ja, a, seq, strip_kb=512, lanes=0, spine=-1, info=None):
    L = fa.lib()
    L.fasp_hip_seq_schedule_selftest.restype = C.c_double
    L.fasp_hip_seq_schedule_selftest.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    A, keep = T.as_csr(ia, ja, a)
    seq = np.ascontiguousarray(seq, dtype=np.int32)
    buf = (C.c_int * 8)()
    r = L.fasp_hip_seq_schedule_selftest(C.byref(A), seq.ctypes.data_as(C.POINTER(C.c_int)), len(seq), strip_kb, lanes, spine, buf)
    if info is not None:
        info.update(lanes=buf[0], rounds=buf[1], spine=buf[2], virtual=buf[3], strips=buf[4], chunks=buf[5])
    return r


@pytest.mark.parametrize("strip_kb", [16, 512])
@pytest.mark.parametrize("lanes", [0, 4])
def test_schedule_reproduces_the_sequential_sweep_on_p7(strip_kb, lanes):
    n = 20
    ia, ja, a, f, ue = fa.poisson7pt(n)
    m = len(f)
    i = np.arange(m)
    red = ((i % n) + (i // n) % n + i // (n * n)) % 2 == 0
    for seq in (i, i[::-1], i[::2], i[red], i[~red][::-1], i[:0]):   # ascending, descending, a subset, red / black (no lower entries), empty
        r = _selftest(ia, ja, a, seq, strip_kb, lanes)
        assert 0.0 <= r <= 1e-13, r


def test_schedule_on_every_level_of_a_hierarchy_with_cf_sweeps():
    """Levels 1.. of a classical hierarchy: rows of 15-60 entries (several lanes per row), unsorted columns, C-row and F-row sweeps."""
    ia, ja, a, f, ue = fa.poisson7pt(24)
    H = fa.AMG(ia, ja, a, fa.param_amg_init(), host_only=True)
    for lev in range(H.num_levels - 1):
        r, c, lia, lja, lv = H.matrix(lev, 0)
        cf = H.cfmark(lev) if hasattr(H, "cfmark") else None
        idx = np.arange(r)
        seqs = [idx, idx[::-1]]
        if cf is not None and len(cf) == r:
            seqs += [idx[cf == 1], idx[cf != 1]]
        for seq in seqs:
            for kb in (16, 512):
                for spine in (0, 1, 2):   # never / where the schedule chooses it (deep levels, >= 8 lanes) / wherever a row has two lanes
                    res = _selftest(lia, lja, lv, seq, kb, 0, spine)
                    assert 0.0 <= res <= 1e-12, (lev, kb, spine, res)
    H.close()


def test_schedule_with_virtual_rows_and_many_lanes():
    """A banded matrix whose rows couple to the 600 rows before them: 64 lanes per row, eight rounds, and more entries than a work
    item holds: the oldest go to VIRTUAL ROWS (seq_sched.h) -- sums of products formed by other waves, read as one operand each;
    strips of 16 KB: one row per chunk, a few rows per strip, every row reads ghosts.  Forced down to 4 and 16 lanes per row a
    row has up to 19 virtual rows."""
    import scipy.sparse as sp
    n, bw = 1500, 600
    offs = list(range(-bw, 0)) + list(range(1, bw + 1))
    A = (sp.diags([-1.0 / (2 * bw)] * len(offs), offs, shape=(n, n), format="csr") + 2.0 * sp.identity(n, format="csr")).tocsr()
    A.sort_indices()
    ia, ja, a = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    for kb in (16, 512):
        for seq in (np.arange(n), np.arange(n)[::-1]):
            for lanes, spine in ((0, 0), (0, 1), (16, 0), (16, 2), (4, 0), (4, 2)):   # (0, 1): a chain of 1500 classes of one row -- the schedule chooses the spine form itself
                info = {}
                res = _selftest(ia, ja, a, seq, kb, lanes, spine, info)
                assert 0.0 <= res <= 1e-12, (kb, lanes, spine, res, info)
                assert info["virtual"] > 0 and info["rounds"] == 8 and info["spine"] == (2 if spine else 0), info
                if lanes:
                    assert info["lanes"] == lanes and info["virtual"] > 2 * n, info


def test_a_row_that_reads_more_than_a_strip_holds_is_reported():
    """One row coupled to 25 000 earlier rows: more than the LDS of a workgroup holds -> no split form (-2: the caller falls back
    to whole-row level scheduling, smoothers.hip.h)."""
    import scipy.sparse as sp
    n = 26000
    A = sp.identity(n, format="lil") * 4.0
    A[n - 1, : n - 1] = -1e-4
    A = A.tocsr()
    r = _selftest(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), np.arange(n))
    assert r == -2.0


def test_rows_without_a_usable_diagonal_are_left_alone():
    """ItrSmootherCSR.c leaves a row alone when |a_ii| <= SMALLREAL: a stored zero diagonal, a missing diagonal, a diagonal stored
    twice (the last hit counts) -- in the middle of the dependency chain."""
    import scipy.sparse as sp
    n = 400
    M = (sp.diags([-1.0, -1.0, 4.0, -1.0, -1.0], [-20, -1, 0, 1, 20], shape=(n, n))).tolil()
    M[50, 50] = 0.0
    M = M.tocoo()
    rows, cols, vals = list(M.row), list(M.col), list(M.data)
    keep = [(r, c, v) for r, c, v in zip(rows, cols, vals) if not (r == 120 and c == 120)]   # row 120: no diagonal entry at all
    keep.append((50, 50, 0.0))                                                               # row 50: stored zero diagonal
    keep += [(200, 200, 1.0), (200, 200, 5.0)]                                               # row 200: two more hits, the last one counts
    keep.sort(key=lambda t: t[0])
    ia = np.zeros(n + 1, np.int32)
    for r, c, v in keep: ia[r + 1] += 1
    ia = np.cumsum(ia).astype(np.int32)
    ja = np.array([c for r, c, v in keep], np.int32); a = np.array([v for r, c, v in keep])
    for seq in (np.arange(n), np.arange(n)[::-1]):
        for kb in (16, 512):
            for lanes, spine in ((0, -1), (8, 2)):
                res = _selftest(ia, ja, a, seq, kb, lanes, spine)
                assert 0.0 <= res <= 1e-12, (kb, lanes, spine, res)
