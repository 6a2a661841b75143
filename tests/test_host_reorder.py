"""Host side of the brick renumbering of the uncoded mid levels (csrc/reorder.cpp; no GPU): the order is a permutation that moves a row
by less than a chunk, its balls are compact (64 consecutive new rows touch far fewer distinct columns than 64 consecutive old ones on a
coarse level in C-point order), and a permuted operator is the same operator: P A P^T, every row's entries in their storage order."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

import faspsolver_amd as fa
from faspsolver_amd import _types as T


def _order(ia, ja, a, chunk):
    L = fa.lib()
    A, keep = T.as_csr(ia, ja, a)
    order = np.zeros(len(ia) - 1, dtype=np.int32)
    st = L.fasp_hip_cluster_order(C.byref(A), chunk, order.ctypes.data_as(C.POINTER(C.c_int)))
    assert st == 0
    return order


def _permute(ia, ja, a, rperm, cinv):
    L = fa.lib()
    A, keep = T.as_csr(ia, ja, a)
    ia2 = np.zeros_like(ia); ja2 = np.zeros_like(ja); a2 = np.zeros_like(a)
    p = lambda v: v.ctypes.data_as(C.POINTER(C.c_int)) if v is not None else None
    st = L.fasp_hip_permute_csr(C.byref(A), p(rperm), p(cinv), p(ia2), p(ja2), a2.ctypes.data_as(C.POINTER(C.c_double)))
    assert st == 0
    return ia2, ja2, a2


def _distinct_columns_per_tile(ia, ja):
    n = len(ia) - 1
    return np.mean([len(np.unique(ja[ia[t]:ia[min(n, t + 64)]])) for t in range(0, n - 63, 64)])


def test_cluster_order_and_permutation_on_a_coarse_level():
    ia, ja, a, f, ue = fa.poisson7pt(32)
    H = fa.AMG(ia, ja, a, fa.param_amg_init(), host_only=True)
    r, c, lia, lja, lv = H.matrix(2, 0)
    H.close()
    assert r == c and r > 1000
    for chunk in (4096, 262144):
        order = _order(lia, lja, lv, chunk)
        assert np.array_equal(np.sort(order), np.arange(r))                      # a permutation
        assert np.all(np.abs(order - np.arange(r)) < max(chunk, 64))             # a row moves inside its chunk
        inv = np.empty(r, dtype=np.int32); inv[order] = np.arange(r, dtype=np.int32)
        pia, pja, pv = _permute(lia, lja, lv, order, inv)
        A = sp.csr_matrix((lv, lja, lia), shape=(r, r))
        B = sp.csr_matrix((pv, pja, pia), shape=(r, r))
        assert (B - A[order][:, order]).nnz == 0                                  # the same operator in the new numbering
        for k in (0, 17, r // 2, r - 1):                                          # every row: the same entries in the same storage order
            i = order[k]
            assert np.array_equal(pv[pia[k]:pia[k + 1]], lv[lia[i]:lia[i + 1]])
            assert np.array_equal(order[pja[pia[k]:pia[k + 1]]], lja[lia[i]:lia[i + 1]])
        # compact balls: what 64 consecutive rows reach
        assert _distinct_columns_per_tile(pia, pja) < 0.75 * _distinct_columns_per_tile(lia, lja)


def test_rectangular_operators_take_one_sided_permutations():
    ia, ja, a, f, ue = fa.poisson7pt(20)
    H = fa.AMG(ia, ja, a, fa.param_amg_init(), host_only=True)
    r, c, pia, pja, pv = H.matrix(1, 1)      # P of level 1: rows level 1, columns level 2
    H.close()
    rng = np.random.default_rng(3)
    rperm = rng.permutation(r).astype(np.int32)
    cperm = rng.permutation(c).astype(np.int32)
    cinv = np.empty(c, dtype=np.int32); cinv[cperm] = np.arange(c, dtype=np.int32)
    P = sp.csr_matrix((pv, pja, pia), shape=(r, c))
    for rp, ci, ref in ((rperm, None, P[rperm]), (None, cinv, P[:, cperm]), (rperm, cinv, P[rperm][:, cperm])):
        qia, qja, qv = _permute(pia, pja, pv, rp, ci)
        assert (sp.csr_matrix((qv, qja, qia), shape=(r, c)) - ref).nnz == 0
