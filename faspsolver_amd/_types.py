"""ctypes mirrors of include/fasp_hip.h (layout == serial reference headers).

Reference: base/include/fasp.h:151 (dCSRmat), :354 (dvector), :368 (ivector),
:386 (ITS_param), :455 (AMG_param); constants base/include/fasp_const.h.
Pure Python: importing this module loads no native library.
"""
import ctypes as C

import numpy as np

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)

# --- constants (fasp_const.h) -------------------------------------------------
FASP_SUCCESS = 0
ERROR_INPUT_PAR = -13
ERROR_MISC = -19
ERROR_ALLOC_MEM = -20
ERROR_AMG_INTERP_TYPE = -30
ERROR_AMG_SMOOTH_TYPE = -31
ERROR_AMG_COARSE_TYPE = -32
ERROR_SOLVER_TYPE = -40
ERROR_SOLVER_PRECTYPE = -41
ERROR_SOLVER_STAG = -42
ERROR_SOLVER_SOLSTAG = -43
ERROR_SOLVER_TOLSMALL = -44
ERROR_SOLVER_MAXIT = -48
ERROR_UNKNOWN = -99

PRINT_NONE, PRINT_MIN, PRINT_SOME, PRINT_MORE, PRINT_MOST, PRINT_ALL = 0, 1, 2, 4, 8, 10
SOLVER_DEFAULT, SOLVER_CG, SOLVER_VGMRES, SOLVER_VFGMRES = 0, 1, 5, 6
SOLVER_BiCGstab = 2
SOLVER_GMRES = 4
SOLVER_MinRes, SOLVER_GCG, SOLVER_GCR = 3, 7, 8
STOP_REL_RES, STOP_REL_PRECRES, STOP_MOD_REL_RES = 1, 2, 3
PREC_NULL, PREC_DIAG, PREC_AMG, PREC_FMG = 0, 1, 2, 3
CLASSIC_AMG, SA_AMG, UA_AMG = 1, 2, 3
V_CYCLE, W_CYCLE, AMLI_CYCLE, NL_AMLI_CYCLE, VW_CYCLE, WV_CYCLE = 1, 2, 3, 4, 12, 21
SMOOTHER_JACOBI, SMOOTHER_GS, SMOOTHER_SGS, SMOOTHER_CG, SMOOTHER_SOR = 1, 2, 3, 4, 5
SMOOTHER_SSOR, SMOOTHER_GSOR, SMOOTHER_SGSOR, SMOOTHER_POLY, SMOOTHER_L1DIAG = 6, 7, 8, 9, 10
SMOOTHER_JACOBIF, SMOOTHER_GSF = 11, 12
COARSE_RS, COARSE_RSP, COARSE_CR, COARSE_AC, COARSE_MIS = 1, 2, 3, 4, 5
INTERP_DIR, INTERP_STD, INTERP_ENG, INTERP_RDC, INTERP_EXT = 1, 2, 3, 4, 6
NO_ORDER, CF_ORDER = 0, 1


class dCSRmat(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("nnz", C.c_int),
                ("IA", c_int_p), ("JA", c_int_p), ("val", c_double_p)]


class dBSRmat(C.Structure):
    _fields_ = [("ROW", C.c_int), ("COL", C.c_int), ("NNZ", C.c_int), ("nb", C.c_int),
                ("storage_manner", C.c_int), ("val", c_double_p), ("IA", c_int_p), ("JA", c_int_p)]


class dvector(C.Structure):
    _fields_ = [("row", C.c_int), ("val", c_double_p)]


class ivector(C.Structure):
    _fields_ = [("row", C.c_int), ("val", c_int_p)]


class ITS_param(C.Structure):
    _fields_ = [("print_level", C.c_short), ("itsolver_type", C.c_short),
                ("decoup_type", C.c_short), ("precond_type", C.c_short),
                ("stop_type", C.c_short), ("restart", C.c_int), ("maxit", C.c_int),
                ("tol", C.c_double), ("abstol", C.c_double)]


class AMG_param(C.Structure):
    _fields_ = [
        ("AMG_type", C.c_short), ("print_level", C.c_short), ("maxit", C.c_int),
        ("tol", C.c_double), ("max_levels", C.c_short), ("coarse_dof", C.c_int),
        ("cycle_type", C.c_short), ("quality_bound", C.c_double), ("smoother", C.c_short),
        ("smooth_order", C.c_short), ("presmooth_iter", C.c_short),
        ("postsmooth_iter", C.c_short), ("relaxation", C.c_double),
        ("polynomial_degree", C.c_short), ("coarse_solver", C.c_short),
        ("coarse_scaling", C.c_short), ("amli_degree", C.c_short),
        ("amli_coef", c_double_p), ("nl_amli_krylov_type", C.c_short),
        ("coarsening_type", C.c_short), ("aggregation_type", C.c_short),
        ("aggregation_norm_type", C.c_short), ("interpolation_type", C.c_short),
        ("strong_threshold", C.c_double), ("max_row_sum", C.c_double),
        ("truncation_threshold", C.c_double), ("aggressive_level", C.c_int),
        ("aggressive_path", C.c_int), ("pair_number", C.c_int),
        ("strong_coupled", C.c_double), ("max_aggregation", C.c_int),
        ("tentative_smooth", C.c_double), ("smooth_filter", C.c_short),
        ("smooth_restriction", C.c_short), ("ILU_levels", C.c_short),
        ("ILU_type", C.c_short), ("ILU_lfil", C.c_int), ("ILU_droptol", C.c_double),
        ("ILU_relax", C.c_double), ("ILU_permtol", C.c_double), ("SWZ_levels", C.c_int),
        ("SWZ_mmsize", C.c_int), ("SWZ_maxlvl", C.c_int), ("SWZ_type", C.c_int),
        ("SWZ_blksolver", C.c_int), ("theta", C.c_double)]


class fasp_hip_stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("nhist", C.c_int), ("relres", C.c_double),
                ("absres", C.c_double), ("normr0", C.c_double),
                ("solve_seconds", C.c_double), ("upload_seconds", C.c_double),
                ("spmv_ms", C.c_double), ("spmv_launches", C.c_longlong),
                ("coarse_iters", C.c_longlong), ("vcycles", C.c_longlong)]


# --- numpy <-> struct helpers ---------------------------------------------------
def as_csr(ia, ja, val, ncol=None):
    """Wrap numpy arrays in a dCSRmat.  Returns (struct, keepalive tuple)."""
    ia = np.ascontiguousarray(ia, dtype=np.int32)
    ja = np.ascontiguousarray(ja, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = ia.shape[0] - 1
    A = dCSRmat(n, n if ncol is None else ncol, int(ia[-1]),
                ia.ctypes.data_as(c_int_p), ja.ctypes.data_as(c_int_p),
                val.ctypes.data_as(c_double_p))
    return A, (ia, ja, val)


def as_bsr(ia, ja, val, nb, ncol=None):
    """Wrap numpy arrays in a dBSRmat (row-major nb x nb blocks).  Returns (struct, keepalive)."""
    ia = np.ascontiguousarray(ia, dtype=np.int32)
    ja = np.ascontiguousarray(ja, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64).reshape(-1)
    n = ia.shape[0] - 1
    A = dBSRmat(n, n if ncol is None else ncol, int(ia[-1]), nb, 0, val.ctypes.data_as(c_double_p),
                ia.ctypes.data_as(c_int_p), ja.ctypes.data_as(c_int_p))
    return A, (ia, ja, val)


def as_vec(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return dvector(x.shape[0], x.ctypes.data_as(c_double_p)), x


def csr_arrays(A, copy=True):
    """dCSRmat view -> (ia, ja, val) numpy arrays."""
    n, nnz = A.row, A.nnz
    ia = np.ctypeslib.as_array(A.IA, shape=(n + 1,))
    ja = np.ctypeslib.as_array(A.JA, shape=(max(nnz, 1),))[:nnz]
    val = np.ctypeslib.as_array(A.val, shape=(max(nnz, 1),))[:nnz]
    if copy:
        return ia.copy(), ja.copy(), val.copy()
    return ia, ja, val


def dp(x):
    return x.ctypes.data_as(c_double_p)


def ip(x):
    return x.ctypes.data_as(c_int_p)


PRECOND_FCT = C.CFUNCTYPE(None, c_double_p, c_double_p, C.c_void_p)


class precond(C.Structure):
    """fasp.h:1095-1103: preconditioner data + action z = B r."""
    _fields_ = [("data", C.c_void_p), ("fct", PRECOND_FCT)]
