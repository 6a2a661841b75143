"""Smoothed-aggregation AMG (config 5: VFGMRES(30) + SA-AMG W-cycle).
CPU part: the product's host setup == the oracle == the reference, bit for bit.
GPU part: the device solve against the oracle."""
import ctypes as C

import numpy as np
import pytest

from _libs import DATA, OrcAMG, T, default_params, oracle, orc_solve, poisson7pt, read_csr, read_vec, ref, ref_solve


class OrcSA(OrcAMG):
    def __init__(self, A, param):
        o = oracle()
        o.orc_amg_setup_sa.argtypes = [C.c_void_p, C.POINTER(T.dCSRmat), C.POINTER(T.AMG_param)]
        self.lib = o
        self.buf = C.create_string_buffer(o.orc_sizeof_amg())
        self.status = o.orc_amg_setup_sa(self.buf, C.byref(A), C.byref(param))
        self.num_levels = C.cast(self.buf, T.c_int_p)[0]


def _sa(p): p.AMG_type = T.SA_AMG; p.smoother = T.SMOOTHER_JACOBI
def _sa_nofilter(p): _sa(p); p.smooth_filter = 0
def _sa_tight(p): _sa(p); p.strong_coupled = 0.25; p.max_aggregation = 9


MATS = {"p7_10": lambda: poisson7pt(10)[:3], "p7_20": lambda: poisson7pt(20)[:3],
        "fe": lambda: read_csr(DATA + "/csrmat_FE.dat")}


@pytest.mark.parametrize("mat", list(MATS))
@pytest.mark.parametrize("mod", [_sa, _sa_nofilter, _sa_tight], ids=["default", "nofilter", "tight"])
def test_sa_hierarchy_product_oracle_reference(fa, mat, mod):
    ia, ja, a = MATS[mat]()
    p1 = default_params()[1]; mod(p1)
    p2 = fa.param_amg_init(); mod(p2)
    A, keep = T.as_csr(ia, ja, a)
    O = OrcSA(A, p1)
    P = fa.AMG(ia, ja, a, p2, host_only=True)
    R = ref()
    hr = None
    if R is not None:
        p3 = default_params()[1]; mod(p3)
        hr = R.ref_amg_setup_rs(C.byref(A), C.byref(p3))  # the shim dispatches on AMG_type
        assert R.ref_amg_num_levels(hr) == O.num_levels
    assert O.num_levels == P.num_levels >= 2
    for l in range(O.num_levels):
        for which, nm in ((0, "A"), (1, "P"), (2, "R")):
            if which and l == O.num_levels - 1:
                continue
            mine = T.csr_arrays(getattr(O.level(l), nm))
            prod = P.matrix(l, which)[2:]
            for x, y in zip(mine, prod):
                assert np.array_equal(x, y)
            if hr is not None:
                v = T.dCSRmat(); R.ref_amg_get_matrix(hr, l, which, C.byref(v))
                for x, y in zip(mine, T.csr_arrays(v)):
                    assert np.array_equal(x, y)
    assert bytes(p1) == bytes(p2)
    O.free(); P.close()


def _c5(itp, amgp):
    itp.tol = 1e-8; itp.itsolver_type = T.SOLVER_VFGMRES; itp.restart = 30
    amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE; amgp.smoother = T.SMOOTHER_JACOBI
    amgp.relaxation = 0.6667


@pytest.mark.ref
@pytest.mark.parametrize("n", [12, 24])
def test_sa_solves_bit_exact_vs_reference(n):
    if ref() is None:
        pytest.skip("oracle/_ref not available")
    ia, ja, a, f, ue = poisson7pt(n)
    i1, a1 = default_params(); _c5(i1, a1)
    i2, a2 = default_params(); _c5(i2, a2)
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, i1, a1)
    s2, x2, h2 = ref_solve(ia, ja, a, f, i2, a2)
    assert s1 == s2 and np.array_equal(x1, x2)


def test_aniso_generator(fa):
    o = oracle()
    o.orc_aniso27pt.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(T.dCSRmat), C.POINTER(T.dvector)]
    A = T.dCSRmat(); b = T.dvector()
    assert o.orc_aniso27pt(7, 1.0, 1.0, 0.01, C.byref(A), C.byref(b)) == 0
    i1, j1, a1 = T.csr_arrays(A)
    i2, j2, a2, f2 = fa.aniso27pt(7)
    assert np.array_equal(i1, i2) and np.array_equal(j1, j2) and np.array_equal(a1, a2)
    assert len(a2) == (3 * 7 - 2) ** 3  # 27-point stencil clipped at the boundary
    import scipy.sparse as sp
    M = sp.csr_matrix((a2, j2, i2))
    assert abs(M - M.T).max() == 0.0
    assert np.linalg.eigvalsh(M.toarray()).min() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["p7_16", "p7_32", "aniso_20"])
def test_config5_vfgmres_sa_wcycle_gpu(gpu, case):
    if case.startswith("p7"):
        ia, ja, a, f, ue = poisson7pt(int(case.split("_")[1]))
    else:
        ia, ja, a, f = gpu.aniso27pt(int(case.split("_")[1]))
    itp, amgp = default_params(); _c5(itp, amgp)
    itp2, amgp2 = default_params(); _c5(itp2, amgp2)
    s_ref, x_ref, h_ref, rr_ref = orc_solve(ia, ja, a, f, itp, amgp)
    H = gpu.AMG(ia, ja, a, amgp2)
    s, x, h, stats = H.solve(f, itp2)
    H.close()
    assert s == s_ref
    assert abs(stats.relres - rr_ref) <= 1e-6 * rr_ref + 1e-15
    assert np.max(np.abs(x - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
