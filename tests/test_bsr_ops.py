"""Block (BSR) operators of config 3: dBSRmat SpMV / aAxpy / inverse diagonal blocks / block
Jacobi.  CPU: oracle vs the reference build (bit-exact).  GPU: HIP kernels vs the oracle --
bit-exact too, the kernel evaluates every block row in the reference's order."""
import ctypes as C

import numpy as np
import pytest

from _libs import DATA, T, oracle, poisson7pt_bsr, read_bsr, read_vec, ref


def _cases():
    return {"spe01": lambda: read_bsr(DATA + "/bsrmat_SPE01.dat"),
            "p7x3_8": lambda: poisson7pt_bsr(8),
            "p7x2_6": lambda: poisson7pt_bsr(6, np.array([[2.0, -1.0], [0.5, 3.0]])),
            "p7x1_7": lambda: poisson7pt_bsr(7, np.array([[1.5]])),
            "p7x5_5": lambda: poisson7pt_bsr(5, np.arange(25, dtype=float).reshape(5, 5) / 7 + np.eye(5) * 9)}


def _orc_setup():
    o = oracle()
    o.orc_bsr_mxv.argtypes = [C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    o.orc_bsr_aAxpy.argtypes = [C.c_double, C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    o.orc_bsr_getdiaginv.argtypes = [C.POINTER(T.dBSRmat)]
    o.orc_bsr_getdiaginv.restype = T.c_double_p
    o.orc_bsr_jacobi1.argtypes = [C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p, T.c_double_p]
    o.orc_free.argtypes = [C.c_void_p]
    return o


@pytest.mark.ref
@pytest.mark.parametrize("case", ["spe01", "p7x3_8", "p7x2_6", "p7x1_7", "p7x5_5"])
def test_oracle_bsr_ops_vs_reference(case):
    R = ref()
    if R is None:
        pytest.skip("oracle/_ref not available")
    o = _orc_setup()
    ia, ja, val, nb = _cases()[case]()
    A, keep = T.as_bsr(ia, ja, val, nb)
    n = A.ROW * nb
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n); y0 = rng.standard_normal(n)
    R.fasp_blas_dbsr_mxv.argtypes = [C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    R.fasp_blas_dbsr_aAxpy.argtypes = [C.c_double, C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
    y1 = np.zeros(n); y2 = np.ones(n)
    o.orc_bsr_mxv(C.byref(A), T.dp(x), T.dp(y1)); R.fasp_blas_dbsr_mxv(C.byref(A), T.dp(x), T.dp(y2))
    assert np.array_equal(y1, y2)
    for alpha in (1.0, -1.0, 0.3):
        y1 = y0.copy(); y2 = y0.copy()
        o.orc_bsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y1)); R.fasp_blas_dbsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y2))
        assert np.array_equal(y1, y2)
    if nb <= 3:
        R.fasp_dbsr_getdiaginv.argtypes = [C.POINTER(T.dBSRmat)]
        R.fasp_dbsr_getdiaginv.restype = T.dvector
        dr = R.fasp_dbsr_getdiaginv(C.byref(A))
        d_ref = np.ctypeslib.as_array(dr.val, (dr.row,)).copy()
        dp_ = o.orc_bsr_getdiaginv(C.byref(A))
        d_orc = np.ctypeslib.as_array(dp_, (A.ROW * nb * nb,)).copy()
        o.orc_free(dp_)
        assert np.array_equal(d_ref, d_orc)
        R.fasp_smoother_dbsr_jacobi1.argtypes = [C.POINTER(T.dBSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), T.c_double_p]
        b = rng.standard_normal(n)
        u1 = x.copy(); u2 = x.copy()
        o.orc_bsr_jacobi1(C.byref(A), T.dp(b), T.dp(u1), T.dp(d_orc))
        bv = T.dvector(n, T.dp(b)); uv = T.dvector(n, T.dp(u2))
        R.fasp_smoother_dbsr_jacobi1(C.byref(A), C.byref(bv), C.byref(uv), T.dp(d_ref))
        assert np.array_equal(u1, u2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["spe01", "p7x3_8", "p7x2_6", "p7x1_7", "p7x5_5"])
def test_gpu_bsr_ops_bit_exact(gpu, case):
    o = _orc_setup()
    L = gpu.lib()
    ia, ja, val, nb = _cases()[case]()
    A, keep = T.as_bsr(ia, ja, val, nb)
    n = A.ROW * nb
    rng = np.random.default_rng(7)
    x = rng.standard_normal(n); y0 = rng.standard_normal(n)
    y1 = np.zeros(n); y2 = np.ones(n)
    o.orc_bsr_mxv(C.byref(A), T.dp(x), T.dp(y1)); L.fasp_blas_dbsr_mxv(C.byref(A), T.dp(x), T.dp(y2))
    assert np.array_equal(y1, y2)
    for alpha in (1.0, -1.0, 0.3):
        y1 = y0.copy(); y2 = y0.copy()
        o.orc_bsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y1)); L.fasp_blas_dbsr_aAxpy(alpha, C.byref(A), T.dp(x), T.dp(y2))
        assert np.array_equal(y1, y2)
    if nb <= 3:
        dv = L.fasp_dbsr_getdiaginv(C.byref(A))
        d = np.ctypeslib.as_array(dv.val, (dv.row,)).copy()
        dp_ = o.orc_bsr_getdiaginv(C.byref(A))
        assert np.array_equal(d, np.ctypeslib.as_array(dp_, (A.ROW * nb * nb,)))
        o.orc_free(dp_)
        b = rng.standard_normal(n)
        u1 = x.copy(); u2 = x.copy()
        o.orc_bsr_jacobi1(C.byref(A), T.dp(b), T.dp(u1), T.dp(d))
        bv = T.dvector(n, T.dp(b)); uv = T.dvector(n, T.dp(u2))
        L.fasp_smoother_dbsr_jacobi1(C.byref(A), C.byref(bv), C.byref(uv), T.dp(d))
        assert np.array_equal(u1, u2)


def test_spe01_shape():
    ia, ja, val, nb = read_bsr(DATA + "/bsrmat_SPE01.dat")
    assert (len(ia) - 1, len(ja), nb) == (302, 1788, 3)
    assert len(read_vec(DATA + "/rhs_SPE01.dat")) == 906
