"""Unsmoothed-aggregation AMG on scalar matrices (PreAMGSetupUA.c:55; VMB aggregation and the reference's
default, symmetric pairwise matching with 1-3 passes, PreAMGAggregationUA.inl:363):
hierarchies bit-identical between product host setup, oracle and the compiled reference; solves on the device."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA, OrcAMG, default_params, have_ref, oracle, orc_solve, poisson7pt, read_csr, read_vec, ref, ref_solve

needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def ua_vmb(p):
    p.AMG_type = T.UA_AMG; p.aggregation_type = 2; p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667


def ua_pair(p):   # fasp_param_amg_init's aggregation_type (PAIRWISE), pair_number 2, quality_bound 10
    p.AMG_type = T.UA_AMG; p.smoother = T.SMOOTHER_JACOBI; p.relaxation = 0.6667


def ua_pair1(p): ua_pair(p); p.pair_number = 1
def ua_pair3(p): ua_pair(p); p.pair_number = 3; p.quality_bound = 8.0


UAS = [ua_vmb, ua_pair, ua_pair1, ua_pair3]
UA_IDS = ["vmb", "pairwise", "pairwise1", "pairwise3"]


def mats():
    out = {}
    for n in (10, 16):
        ia, ja, a, f, ue = poisson7pt(n)
        out[f"p7_{n}"] = (ia, ja, a, f)
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat")
    out["fe"] = (ia, ja, a, read_vec(DATA + "/rhs_FE.dat"))
    return out


MATS = mats()


def orc_hierarchy(ia, ja, a, p):
    o = oracle()
    o.orc_amg_setup_ua.argtypes = [C.c_void_p, C.POINTER(T.dCSRmat), C.POINTER(T.AMG_param)]
    A, keep = T.as_csr(ia, ja, a)
    H = OrcAMG.__new__(OrcAMG)
    H.lib = o
    H.buf = C.create_string_buffer(o.orc_sizeof_amg())
    H.status = o.orc_amg_setup_ua(H.buf, C.byref(A), C.byref(p))
    H.num_levels = C.cast(H.buf, T.c_int_p)[0]
    return H


@pytest.mark.parametrize("ua", UAS, ids=UA_IDS)
@pytest.mark.parametrize("name", list(MATS))
def test_ua_hierarchy_product_equals_oracle(name, ua):
    ia, ja, a, f = MATS[name]
    _, p1 = default_params(); ua(p1); _, p2 = default_params(); ua(p2)
    H = orc_hierarchy(ia, ja, a, p1)
    G = fa.AMG(ia, ja, a, p2, host_only=True)
    assert G.num_levels == H.num_levels and H.num_levels >= 2
    assert bytes(p1) == bytes(p2)  # strong_coupled adapted identically
    for l in range(H.num_levels):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == H.num_levels - 1:
                continue
            m = getattr(H.level(l), nm)
            r, c, i2, j2, v = G.matrix(l, which)
            assert (m.row, m.col, m.nnz) == (r, c, len(v))
            mi, mj, mv = T.csr_arrays(m)
            assert np.array_equal(mi, i2) and np.array_equal(mj, j2) and np.array_equal(mv, v)
    G.close(); H.free()


@needs_ref
@pytest.mark.parametrize("ua", UAS, ids=UA_IDS)
@pytest.mark.parametrize("name", list(MATS))
def test_ua_hierarchy_oracle_equals_reference(name, ua):
    ia, ja, a, f = MATS[name]
    R = ref()
    _, p1 = default_params(); ua(p1); _, p2 = default_params(); ua(p2)
    H = orc_hierarchy(ia, ja, a, p1)
    A, keep = T.as_csr(ia, ja, a)
    h = R.ref_amg_setup_rs(C.byref(A), C.byref(p2))  # dispatches on AMG_type
    assert R.ref_amg_num_levels(h) == H.num_levels
    for l in range(H.num_levels):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == H.num_levels - 1:
                continue
            v = T.dCSRmat(); R.ref_amg_get_matrix(h, l, which, C.byref(v))
            m = getattr(H.level(l), nm)
            assert (m.row, m.col, m.nnz) == (v.row, v.col, v.nnz)
            assert all(np.array_equal(x, y) for x, y in zip(T.csr_arrays(m), T.csr_arrays(v)))
    assert bytes(p1) == bytes(p2)
    H.free()


def _pcg(i, p): i.tol = 1e-8
def _vfg_w(i, p): i.tol = 1e-8; i.itsolver_type = 6; i.restart = 30; p.cycle_type = 2


@needs_ref
@pytest.mark.parametrize("mod", [_pcg, _vfg_w], ids=["pcg_V", "vfgmres_W"])
@pytest.mark.parametrize("ua", UAS, ids=UA_IDS)
@pytest.mark.parametrize("name", list(MATS))
def test_ua_solves_oracle_equals_reference(name, mod, ua):
    ia, ja, a, f = MATS[name]
    i1, a1 = default_params(); ua(a1); mod(i1, a1)
    i2, a2 = default_params(); ua(a2); mod(i2, a2)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1)
    s2, x2, h2 = ref_solve(ia, ja, a, f, i2, a2)
    assert s1 == s2 and s1 > 0
    assert np.array_equal(x1, x2)


def test_ua_unsymmetric_pairwise_is_refused():
    ia, ja, a, f = MATS["p7_10"]
    itp, amgp = default_params(); amgp.AMG_type = T.UA_AMG; amgp.smoother = T.SMOOTHER_JACOBI; amgp.aggregation_type = 3  # NPAIR
    x = np.zeros(len(f))
    assert fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, itp, amgp) == T.ERROR_INPUT_PAR


@pytest.mark.gpu
@pytest.mark.parametrize("mod", [_pcg, _vfg_w], ids=["pcg_V", "vfgmres_W"])
@pytest.mark.parametrize("ua", UAS, ids=UA_IDS)
@pytest.mark.parametrize("n", [16, 32])
def test_gpu_ua_solve_matches_oracle(n, mod, ua):
    ia, ja, a, f, ue = poisson7pt(n)
    i1, a1 = default_params(); ua(a1); mod(i1, a1)
    i2, a2 = default_params(); ua(a2); mod(i2, a2)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1)
    x2 = np.zeros(len(f))
    s2 = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x2, i2, a2)
    assert s2 == s1
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()
