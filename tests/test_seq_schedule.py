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


# ---- the chain form (csrc/seq_chain.hip.h, round 5): blocked substitution with the dependency chain inside one wavefront ----
def _chain_selftest(ia, ja, a, seq, n1=0, form=1, w=1.0, info=None, out=None):
    L = fa.lib()
    L.fasp_hip_seq_chain_selftest.restype = C.c_double
    L.fasp_hip_seq_chain_selftest.argtypes = [C.POINTER(T.dCSRmat), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    A, keep = T.as_csr(ia, ja, a)
    seq = np.ascontiguousarray(seq, dtype=np.int32)
    buf = (C.c_int * 10)()
    r = L.fasp_hip_seq_chain_selftest(C.byref(A), seq.ctypes.data_as(C.POINTER(C.c_int)), len(seq), n1, form, w, buf,
                                      out.ctypes.data_as(C.POINTER(C.c_double)) if out is not None else None)
    if info is not None:
        info.update(blocks=buf[0], n1b=buf[1], rx=buf[2], rg=buf[3], t1_steps=buf[4], t2_steps=buf[5], band=buf[6], t1=buf[7], t2=buf[8], classes=buf[9])
    return r


def _banded(n, bw, fill=1.0, seed=0):
    """Rows couple to the bw rows on either side with probability `fill` (unsorted columns, diagonally dominant)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    offs = list(range(-bw, 0)) + list(range(1, bw + 1))
    A = sp.diags([-1.0 / (2 * bw)] * len(offs), offs, shape=(n, n), format="lil")
    if fill < 1.0:
        A = A.tocoo()
        keep = rng.random(A.nnz) < fill
        A = sp.coo_matrix((A.data[keep] * (0.5 + rng.random(keep.sum())), (A.row[keep], A.col[keep])), shape=(n, n))
    A = (A.tocsr() + 2.0 * sp.identity(n, format="csr")).tocsr()
    ia, ja, a = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    for i in range(n):   # shuffle the entries of every row: the setup's rows are unsorted, the diagonal sits anywhere
        p = rng.permutation(ia[i + 1] - ia[i]) + ia[i]
        ja[ia[i]:ia[i + 1]], a[ia[i]:ia[i + 1]] = ja[p], a[p]
    return ia, ja, a


@pytest.mark.parametrize("form,w", [(0, 1.0), (1, 1.0), (2, 1.1)])
def test_chain_schedule_reproduces_the_sequential_sweep(form, w):
    """Banded matrices of the deep levels' kind (hundreds of lower entries per row, one or two rows per dependency class): every tier is
    populated -- the band (two blocks), tier 1 (n1 blocks, forced small so that tier 2 exists) -- ascending, descending and subset
    sweeps, sizes that are no multiple of 64."""
    for n, bw, fill in ((1500, 600, 1.0), (1000, 700, 0.3), (333, 200, 0.5), (64, 10, 1.0), (65, 64, 1.0), (40, 5, 1.0)):
        ia, ja, a = _banded(n, bw, fill, seed=n)
        idx = np.arange(n)
        for seq in (idx, idx[::-1], idx[idx % 3 != 1]):
            for n1 in (0, 1, 4):
                info = {}
                r = _chain_selftest(ia, ja, a, seq, n1, form, w, info)
                assert 0.0 <= r <= 1e-12, (n, bw, n1, r, info)
                assert info["blocks"] == (len(seq) + 63) // 64 and info["rx"] == 64 * (info["n1b"] + (info["rg"] // 64 - 2) + 3), info   # (rg = 64 (CHAIN_HA + 2))
                if bw * len(seq) // n >= 64 * (info["n1b"] + 3) and len(seq) > 64 * (info["n1b"] + 3):   # (the band in sweep positions reaches beyond tier 1)
                    assert info["t2"] > 0 and info["t1"] > 0 and info["band"] > 0, info


def test_chain_schedule_is_independent_of_the_tier_split():
    """Tier 1 and tier 2 are both per-row chains of fused multiply-adds in column order, but G2 starts from the right-hand side and tier 1
    from zero: the split point is part of a schedule's arithmetic (a regrouping of the same products: differences of an ulp or two), the
    result is the sequential sweep either way."""
    ia, ja, a = _banded(1200, 500, 0.6, seed=5)
    outs = []
    for n1 in (1, 2, 6, 0):
        u = np.zeros(1200)
        assert 0.0 <= _chain_selftest(ia, ja, a, np.arange(1200), n1, 1, 1.0, None, u) <= 1e-12
        outs.append(u)
    for u in outs[1:]:
        assert np.abs(u - outs[0]).max() <= 1e-13 * np.abs(outs[0]).max()


def test_chain_schedule_on_the_deep_levels_of_a_hierarchy():
    """C-row, F-row and natural-order sweeps of the deeper levels of a classical hierarchy (P7(40): levels of 60-250 entries per row)."""
    ia, ja, a, f, ue = fa.poisson7pt(40)
    H = fa.AMG(ia, ja, a, fa.param_amg_init(), host_only=True)
    done = 0
    for lev in range(2, H.num_levels - 1):
        r, c, lia, lja, lv = H.matrix(lev, 0)
        cf = H.cfmark(lev)
        idx = np.arange(r)
        for seq in (idx, idx[::-1], idx[cf == 1], idx[cf != 1]):
            if len(seq) == 0:
                continue
            for form, w in ((1, 1.0), (2, 1.1)):
                info = {}
                res = _chain_selftest(lia, lja, lv, seq, 0, form, w, info)
                assert 0.0 <= res <= 1e-12, (lev, form, res, info)
                done += 1
    assert done >= 16
    H.close()


def test_chain_form_refuses_what_it_cannot_do():
    """A row the reference leaves alone (|a_ii| <= 1e-20): the dataflow form handles it, the chain form says 'does not apply'."""
    ia, ja, a = _banded(300, 100, 1.0)
    a2 = a.copy()
    i = 150
    a2[ia[i]:ia[i + 1]][ja[ia[i]:ia[i + 1]] == i] = 0.0
    assert _chain_selftest(ia, ja, a2, np.arange(300)) == -2.0
    assert 0.0 <= _selftest(ia, ja, a2, np.arange(300)) <= 1e-12
