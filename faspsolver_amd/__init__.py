"""faspsolver_amd -- host-side mirror of the C-ABI in include/fasp_hip.h.

The product is libfasp_hip.so (C host code + hand-written HIP kernels for gfx950,
built by faspsolver_amd/csrc/Makefile).  This module only binds it with ctypes and
mirrors the reference's call signatures (same names, argument meaning and error
behaviour as base/src/SolCSR.c:476, AuxParam.c:431/572, ...), so that parity tests
read like the reference's own drivers.  It never computes anything itself and has no
CPU fallback: if the shared library is missing it raises; if there is no GPU the
library's compute entry points return ERROR_MISC.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _types as T
from ._types import *  # noqa: F401,F403  (constants + struct mirrors)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FASP_HIP_LIB") or os.path.join(_HERE, "libfasp_hip.so")   # (FASP_HIP_LIB: a development build beside the product one, tools/build_variant.sh)
_lib = None

EXPORTS = [  # every symbol include/fasp_hip.h declares
    "fasp_param_amg_init", "fasp_param_solver_init", "fasp_solver_dcsr_krylov_amg",
    "fasp_blas_dcsr_mxv", "fasp_blas_dcsr_aAxpy", "fasp_blas_darray_dotprod",
    "fasp_blas_darray_norm2", "fasp_blas_darray_norminf", "fasp_blas_darray_axpy",
    "fasp_blas_darray_axpby", "fasp_smoother_dcsr_jacobi",
    "fasp_blas_dbsr_mxv", "fasp_blas_dbsr_aAxpy", "fasp_dbsr_getdiaginv", "fasp_smoother_dbsr_jacobi1",
    "fasp_solver_dbsr_pcg", "fasp_solver_dbsr_pbcgs", "fasp_solver_dbsr_pgmres", "fasp_solver_dbsr_pvgmres",
    "fasp_solver_dbsr_pvfgmres", "fasp_hip_bsr_precond_setup", "fasp_hip_bsr_precond_fct", "fasp_hip_bsr_precond_free",
    "fasp_hip_param_input", "fasp_fwrapper_dcsr_krylov_amg_", "fasp_dcsrvec_read2", "fasp_dvec_read",
    "fasp_dbsr_read", "fasp_dcoo_read", "fasp_dcoo_read1", "fasp_dcoo_shift_read", "fasp_dmtx_read", "fasp_dmtxsym_read", "fasp_dvec_write", "fasp_dcsr_write_coo", "fasp_dcsrvec_write2", "fasp_fwrapper_dcsr_amg_", "fasp_fwrapper_dbsr_krylov_amg_", "fasp_hip_free_bsr", "fasp_hip_comm_selftest", "fasp_hip_amg_kernel_info", "fasp_hip_coding_selftest", "fasp_solver_amg", "fasp_solver_famg", "fasp_hip_amg_solve", "fasp_precond_diag", "fasp_precond_dbsr_diag", "fasp_solver_dcsr_itsolver", "fasp_solver_dcsr_krylov", "fasp_solver_dcsr_krylov_diag", "fasp_solver_dbsr_itsolver", "fasp_solver_dbsr_krylov", "fasp_solver_dbsr_krylov_diag", "fasp_hip_mxv_csr", "fasp_hip_mxv_bsr", "fasp_solver_matfree_init", "fasp_solver_pcg", "fasp_solver_pbcgs", "fasp_solver_pgcg", "fasp_solver_pminres", "fasp_solver_pgmres", "fasp_solver_pvgmres", "fasp_solver_pvfgmres", "fasp_solver_itsolver", "fasp_solver_krylov", "fasp_solver_dcsr_pcg", "fasp_solver_dcsr_pminres", "fasp_solver_dcsr_pgcg", "fasp_solver_dcsr_pgcr", "fasp_solver_dcsr_pbcgs", "fasp_solver_dcsr_pgmres", "fasp_solver_dcsr_pvgmres",
    "fasp_solver_dcsr_pvfgmres", "fasp_hip_precond_setup", "fasp_hip_precond_fct", "fasp_hip_precond_free",
    "fasp_hip_time_bsr_mxv", "fasp_solver_dbsr_krylov_amg", "fasp_hip_bsr_amg_create", "fasp_hip_bsr_amg_create_host",
    "fasp_hip_bsr_amg_destroy", "fasp_hip_bsr_amg_num_levels", "fasp_hip_bsr_amg_get_matrix",
    "fasp_hip_bsr_amg_get_diaginv", "fasp_hip_bsr_solve",
    "fasp_hip_set_device", "fasp_hip_device_count", "fasp_hip_device_identity", "fasp_hip_available",
    "fasp_hip_amg_create", "fasp_hip_amg_create_host", "fasp_hip_amg_upload",
    "fasp_hip_amg_destroy", "fasp_hip_amg_num_levels", "fasp_hip_amg_get_matrix",
    "fasp_hip_amg_get_cfmark", "fasp_hip_solve", "fasp_hip_set_rhs", "fasp_hip_set_guess",
    "fasp_hip_solve_resident", "fasp_hip_get_solution", "fasp_hip_device_synchronize",
    "fasp_hip_precond_amg",
    "fasp_hip_poisson7pt", "fasp_hip_aniso27pt", "fasp_hip_free_system", "fasp_hip_time_kernel", "fasp_hip_measure_ceilings", "fasp_hip_amg_publish", "fasp_hip_amg_attach", "fasp_hip_amg_unpublish",
    "fasp_precond_setup", "fasp_precond_amg", "fasp_precond_famg", "fasp_precond_amli", "fasp_precond_namli", "fasp_amg_data_create", "fasp_amg_data_free", "fasp_param_amg_to_prec", "fasp_param_prec_to_amg", "fasp_mem_free", "fasp_mem_calloc", "fasp_dvec_alloc", "fasp_dvec_set", "fasp_dvec_free", "fasp_dvec_create", "fasp_dcsr_create", "fasp_dcsr_free", "fasp_smoother_dcsr_gs", "fasp_smoother_dcsr_sor", "fasp_smoother_dcsr_L1diag",
    "fasp_hip_tune", "fasp_hip_comm_unique_id", "fasp_hip_comm_init", "fasp_hip_comm_finalize",
    "fasp_hip_comm_rank", "fasp_hip_comm_size", "fasp_hip_version", "fasp_hip_comm_init_shm",
    "fasp_hip_dist_plan", "fasp_hip_dist_level_info", "fasp_hip_dist_get_matrix",
    "fasp_hip_dist_get_list",
    "fasp_blas_dcsr_vmv", "fasp_blas_dcsr_mxv_agg", "fasp_blas_dcsr_aAxpy_agg", "fasp_blas_darray_ax",
    "fasp_blas_darray_axpyz", "fasp_blas_darray_norm1", "fasp_darray_cp", "fasp_darray_set", "fasp_dvec_isnan",
    "fasp_hip_time_matrix", "fasp_hip_bsr_dist_info", "fasp_hip_seq_schedule_selftest", "fasp_hip_seq_chain_selftest", "fasp_hip_cluster_order", "fasp_hip_permute_csr", "fasp_hip_comm_init_ipc", "fasp_hip_comm_stats", "fasp_hip_comm_timing", "fasp_hip_estream_selftest",
]


def build(verbose=False):
    """Compile libfasp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)


def lib():
    """The loaded shared library (built on demand).  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)  # RTLD_LOCAL: the reference-named symbols must not interpose other libraries
    P = C.POINTER
    L.fasp_param_amg_init.argtypes = [P(T.AMG_param)]
    L.fasp_param_solver_init.argtypes = [P(T.ITS_param)]
    L.fasp_solver_dcsr_krylov_amg.argtypes = [P(T.dCSRmat), P(T.dvector), P(T.dvector),
                                              P(T.ITS_param), P(T.AMG_param)]
    L.fasp_blas_dcsr_mxv.argtypes = [P(T.dCSRmat), T.c_double_p, T.c_double_p]
    L.fasp_blas_dcsr_aAxpy.argtypes = [C.c_double, P(T.dCSRmat), T.c_double_p, T.c_double_p]
    for f in ("fasp_blas_darray_dotprod", "fasp_blas_darray_norm2", "fasp_blas_darray_norminf"):
        getattr(L, f).restype = C.c_double
    L.fasp_blas_darray_dotprod.argtypes = [C.c_int, T.c_double_p, T.c_double_p]
    L.fasp_blas_darray_norm2.argtypes = [C.c_int, T.c_double_p]
    L.fasp_blas_darray_norminf.argtypes = [C.c_int, T.c_double_p]
    L.fasp_blas_darray_axpy.argtypes = [C.c_int, C.c_double, T.c_double_p, T.c_double_p]
    L.fasp_blas_darray_axpby.argtypes = [C.c_int, C.c_double, T.c_double_p, C.c_double,
                                         T.c_double_p]
    L.fasp_blas_dcsr_vmv.restype = C.c_double
    L.fasp_blas_dcsr_vmv.argtypes = [P(T.dCSRmat), T.c_double_p, T.c_double_p]
    L.fasp_blas_dcsr_mxv_agg.argtypes = [P(T.dCSRmat), T.c_double_p, T.c_double_p]
    L.fasp_blas_dcsr_aAxpy_agg.argtypes = [C.c_double, P(T.dCSRmat), T.c_double_p, T.c_double_p]
    L.fasp_blas_darray_ax.argtypes = [C.c_int, C.c_double, T.c_double_p]
    L.fasp_blas_darray_axpyz.argtypes = [C.c_int, C.c_double, T.c_double_p, T.c_double_p, T.c_double_p]
    L.fasp_blas_darray_norm1.restype = C.c_double
    L.fasp_blas_darray_norm1.argtypes = [C.c_int, T.c_double_p]
    L.fasp_darray_cp.argtypes = [C.c_int, T.c_double_p, T.c_double_p]
    L.fasp_darray_set.argtypes = [C.c_int, T.c_double_p, C.c_double]
    L.fasp_hip_time_matrix.restype = C.c_double
    L.fasp_hip_time_matrix.argtypes = [P(T.dCSRmat), C.c_int, C.c_int, P(C.c_int)]
    L.fasp_dvec_isnan.restype = C.c_short
    L.fasp_dvec_isnan.argtypes = [P(T.dvector)]
    L.fasp_smoother_dcsr_jacobi.argtypes = [P(T.dvector), C.c_int, C.c_int, C.c_int,
                                            P(T.dCSRmat), P(T.dvector), C.c_int, C.c_double]
    L.fasp_blas_dbsr_mxv.argtypes = [P(T.dBSRmat), T.c_double_p, T.c_double_p]
    L.fasp_blas_dbsr_aAxpy.argtypes = [C.c_double, P(T.dBSRmat), T.c_double_p, T.c_double_p]
    L.fasp_dbsr_getdiaginv.argtypes = [P(T.dBSRmat)]
    L.fasp_dbsr_getdiaginv.restype = T.dvector
    L.fasp_smoother_dbsr_jacobi1.argtypes = [P(T.dBSRmat), P(T.dvector), P(T.dvector), T.c_double_p]
    L.fasp_hip_time_bsr_mxv.argtypes = [P(T.dBSRmat), C.c_int]
    L.fasp_hip_time_bsr_mxv.restype = C.c_double
    L.fasp_solver_dcsr_pcg.argtypes = [P(T.dCSRmat), P(T.dvector), P(T.dvector), P(T.precond), C.c_double,
                                       C.c_double, C.c_int, C.c_short, C.c_short]
    L.fasp_solver_dcsr_pbcgs.argtypes = L.fasp_solver_dcsr_pcg.argtypes
    L.fasp_solver_dcsr_pminres.argtypes = L.fasp_solver_dcsr_pcg.argtypes
    L.fasp_solver_dcsr_pgcg.argtypes = L.fasp_solver_dcsr_pcg.argtypes
    L.fasp_solver_dcsr_pvgmres.argtypes = [P(T.dCSRmat), P(T.dvector), P(T.dvector), P(T.precond), C.c_double,
                                           C.c_double, C.c_int, C.c_short, C.c_short, C.c_short]
    L.fasp_solver_dcsr_pvfgmres.argtypes = L.fasp_solver_dcsr_pvgmres.argtypes
    L.fasp_solver_dcsr_pgmres.argtypes = L.fasp_solver_dcsr_pvgmres.argtypes
    L.fasp_solver_dcsr_pgcr.argtypes = L.fasp_solver_dcsr_pvgmres.argtypes
    L.fasp_hip_precond_setup.argtypes = [P(T.dCSRmat), P(T.AMG_param)]
    L.fasp_hip_precond_setup.restype = P(T.precond)
    L.fasp_hip_precond_free.argtypes = [P(T.precond)]
    L.fasp_hip_precond_free.restype = None
    L.fasp_hip_precond_fct.argtypes = [T.c_double_p, T.c_double_p, C.c_void_p]
    L.fasp_hip_precond_fct.restype = None
    L.fasp_solver_dbsr_pcg.argtypes = [P(T.dBSRmat), P(T.dvector), P(T.dvector), P(T.precond), C.c_double,
                                       C.c_double, C.c_int, C.c_short, C.c_short]
    L.fasp_solver_dbsr_pbcgs.argtypes = L.fasp_solver_dbsr_pcg.argtypes
    L.fasp_solver_dbsr_pvgmres.argtypes = [P(T.dBSRmat), P(T.dvector), P(T.dvector), P(T.precond), C.c_double,
                                           C.c_double, C.c_int, C.c_short, C.c_short, C.c_short]
    L.fasp_solver_dbsr_pgmres.argtypes = L.fasp_solver_dbsr_pvgmres.argtypes
    L.fasp_solver_dbsr_pvfgmres.argtypes = L.fasp_solver_dbsr_pvgmres.argtypes
    L.fasp_hip_bsr_precond_setup.argtypes = [P(T.dBSRmat), P(T.AMG_param)]
    L.fasp_hip_bsr_precond_setup.restype = P(T.precond)
    L.fasp_hip_bsr_precond_free.argtypes = [P(T.precond)]
    L.fasp_hip_bsr_precond_free.restype = None
    L.fasp_hip_bsr_precond_fct.argtypes = [T.c_double_p, T.c_double_p, C.c_void_p]
    L.fasp_hip_bsr_precond_fct.restype = None
    L.fasp_dcsrvec_read2.argtypes = [C.c_char_p, C.c_char_p, P(T.dCSRmat), P(T.dvector)]
    L.fasp_dvec_read.argtypes = [C.c_char_p, P(T.dvector)]
    L.fasp_dbsr_read.argtypes = [C.c_char_p, P(T.dBSRmat)]
    L.fasp_hip_free_bsr.argtypes = [P(T.dBSRmat)]
    L.fasp_hip_free_bsr.restype = None
    L.fasp_hip_param_input.argtypes = [C.c_char_p, P(T.ITS_param), P(T.AMG_param)]
    L.fasp_fwrapper_dcsr_krylov_amg_.argtypes = [P(C.c_int), P(C.c_int), T.c_int_p, T.c_int_p, T.c_double_p,
                                                 T.c_double_p, T.c_double_p, P(C.c_double), P(C.c_int), P(C.c_int)]
    L.fasp_fwrapper_dcsr_krylov_amg_.restype = None
    L.fasp_hip_coding_selftest.argtypes = [P(T.dCSRmat), P(C.c_int)]
    L.fasp_hip_amg_kernel_info.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_int), P(C.c_double)]
    L.fasp_solver_amg.argtypes = [P(T.dCSRmat), P(T.dvector), P(T.dvector), P(T.AMG_param)]
    L.fasp_hip_amg_solve.argtypes = [C.c_void_p, P(T.dvector), P(T.dvector), P(T.AMG_param), T.c_double_p,
                                     C.c_int, P(T.fasp_hip_stats)]
    L.fasp_solver_dbsr_krylov_amg.argtypes = [P(T.dBSRmat), P(T.dvector), P(T.dvector), P(T.ITS_param),
                                              P(T.AMG_param)]
    L.fasp_hip_bsr_amg_create.argtypes = [P(C.c_void_p), P(T.dBSRmat), P(T.AMG_param)]
    L.fasp_hip_bsr_amg_create_host.argtypes = [P(C.c_void_p), P(T.dBSRmat), P(T.AMG_param)]
    L.fasp_hip_bsr_amg_destroy.argtypes = [C.c_void_p]
    L.fasp_hip_bsr_amg_destroy.restype = None
    L.fasp_hip_bsr_amg_num_levels.argtypes = [C.c_void_p]
    L.fasp_hip_bsr_amg_get_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int, P(T.dBSRmat)]
    L.fasp_hip_bsr_amg_get_diaginv.argtypes = [C.c_void_p, C.c_int]
    L.fasp_hip_bsr_amg_get_diaginv.restype = T.c_double_p
    L.fasp_hip_bsr_solve.argtypes = [C.c_void_p, P(T.dvector), P(T.dvector), P(T.ITS_param), T.c_double_p,
                                     C.c_int, P(T.fasp_hip_stats)]
    L.fasp_hip_amg_create.argtypes = [P(C.c_void_p), P(T.dCSRmat), P(T.AMG_param)]
    L.fasp_hip_amg_create_host.argtypes = L.fasp_hip_amg_create.argtypes
    L.fasp_hip_amg_upload.argtypes = [C.c_void_p]
    L.fasp_hip_amg_destroy.argtypes = [C.c_void_p]
    L.fasp_hip_amg_destroy.restype = None
    L.fasp_hip_amg_num_levels.argtypes = [C.c_void_p]
    L.fasp_hip_amg_get_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int, P(T.dCSRmat)]
    L.fasp_hip_amg_get_cfmark.argtypes = [C.c_void_p, C.c_int, P(T.ivector)]
    L.fasp_hip_solve.argtypes = [C.c_void_p, P(T.dvector), P(T.dvector), P(T.ITS_param),
                                 T.c_double_p, C.c_int, P(T.fasp_hip_stats)]
    L.fasp_hip_set_rhs.argtypes = [C.c_void_p, P(T.dvector)]
    L.fasp_hip_set_guess.argtypes = [C.c_void_p, P(T.dvector)]
    L.fasp_hip_get_solution.argtypes = [C.c_void_p, P(T.dvector)]
    L.fasp_hip_solve_resident.argtypes = [C.c_void_p, P(T.ITS_param), T.c_double_p, C.c_int,
                                          P(T.fasp_hip_stats)]
    L.fasp_hip_precond_amg.argtypes = [C.c_void_p, T.c_double_p, T.c_double_p]
    L.fasp_hip_poisson7pt.argtypes = [C.c_int, C.c_int, C.c_int, P(T.dCSRmat), P(T.dvector),
                                      P(T.dvector)]
    L.fasp_hip_aniso27pt.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, P(T.dCSRmat), P(T.dvector)]
    L.fasp_hip_free_system.argtypes = [P(T.dCSRmat), P(T.dvector), P(T.dvector)]
    L.fasp_hip_free_system.restype = None
    L.fasp_hip_measure_ceilings.argtypes = [P(C.c_double), C.c_size_t, C.c_int]
    L.fasp_hip_amg_publish.argtypes = [C.c_void_p, C.c_char_p]
    L.fasp_hip_amg_attach.argtypes = [P(C.c_void_p), C.c_char_p]
    L.fasp_hip_amg_unpublish.argtypes = [C.c_char_p]
    L.fasp_hip_time_kernel.restype = C.c_double
    L.fasp_hip_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.fasp_hip_tune.argtypes = [C.c_char_p, C.c_int]
    L.fasp_hip_comm_unique_id.argtypes = [C.c_char_p]
    L.fasp_hip_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p]
    L.fasp_hip_version.restype = C.c_char_p
    L.fasp_hip_comm_init_shm.argtypes = [C.c_int, C.c_int, C.c_char_p]
    L.fasp_hip_dist_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.fasp_hip_dist_level_info.argtypes = [C.c_void_p, C.c_int, T.c_int_p]
    L.fasp_hip_dist_get_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int, P(T.dCSRmat)]
    L.fasp_hip_dist_get_list.argtypes = [C.c_void_p, C.c_int, C.c_int, P(T.ivector)]
    _lib = L
    return L


def available():
    """True when a HIP device is usable by the library."""
    return bool(lib().fasp_hip_available())


# --- parameter constructors (AuxParam.c:431 / :572) ------------------------------
def param_amg_init():
    p = T.AMG_param()
    lib().fasp_param_amg_init(C.byref(p))
    return p


def param_solver_init():
    p = T.ITS_param()
    lib().fasp_param_solver_init(C.byref(p))
    return p


def poisson7pt(nx, ny=None, nz=None):
    """P7 synthetic system (test/src/FdmPoisson.c:439 + :731) -> (ia, ja, a, f, u_exact)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    A = T.dCSRmat(); b = T.dvector(); u = T.dvector()
    st = lib().fasp_hip_poisson7pt(nx, ny, nz, C.byref(A), C.byref(b), C.byref(u))
    if st < 0:
        raise RuntimeError(f"fasp_hip_poisson7pt failed: {st}")
    ia, ja, a = T.csr_arrays(A)
    f = np.ctypeslib.as_array(b.val, (b.row,)).copy()
    ue = np.ctypeslib.as_array(u.val, (u.row,)).copy()
    lib().fasp_hip_free_system(C.byref(A), C.byref(b), C.byref(u))
    return ia, ja, a, f, ue


def poisson7pt_var(n, p7=None):
    """Variable-coefficient twin of P7(n): -div(kappa grad u) = f on the same grid and sparsity pattern,
    kappa = 1 + 0.8 sin(2 pi x) sin(2 pi y) sin(2 pi z) (contrast 9), face coefficients = mean of the two
    node values, Dirichlet boundary.  No two rows repeat, so no lossless matrix coding applies: the
    plain-CSR kernels serve every level.  Entry order = P7's (diagonal first).  -> (ia, ja, a, f)"""
    ia, ja, a, f, _ue = p7 if p7 is not None else poisson7pt(n)
    m = len(f)
    h = 1.0 / (n + 1)
    idx = np.arange(m, dtype=np.int64)
    xi = (idx % n + 1) * h
    yi = ((idx // n) % n + 1) * h
    zi = (idx // (n * n) + 1) * h
    kap = 1.0 + 0.8 * np.sin(2 * np.pi * xi) * np.sin(2 * np.pi * yi) * np.sin(2 * np.pi * zi)
    del xi, yi, zi
    cnt = np.diff(ia)
    rows = np.repeat(idx, cnt)
    off = ja != rows
    a2 = np.where(off, a * 0.5 * (kap[rows] + kap[ja]), 0.0)          # -kappa_face / h^2
    rowsum = np.bincount(rows, weights=a2, minlength=m)                # sum of the off-diagonals (negative)
    diag = -rowsum + (7 - cnt) * kap / (h * h)                         # + the faces that touch the boundary
    a2[ia[:-1]] = diag                                                 # P7 stores the diagonal first
    return ia, ja, a2, f.copy()


def aniso27pt(n, kx=1.0, ky=1.0, kz=0.01):
    """Config-5 synthetic system: Q1 FE, -div(diag(kx,ky,kz) grad u) = 1 -> (ia, ja, a, f)."""
    A = T.dCSRmat(); b = T.dvector()
    st = lib().fasp_hip_aniso27pt(n, kx, ky, kz, C.byref(A), C.byref(b))
    if st < 0:
        raise RuntimeError(f"fasp_hip_aniso27pt failed: {st}")
    ia, ja, a = T.csr_arrays(A)
    f = np.ctypeslib.as_array(b.val, (b.row,)).copy()
    lib().fasp_hip_free_system(C.byref(A), C.byref(b), None)
    return ia, ja, a, f


def solver_dcsr_krylov_amg(ia, ja, a, b, x, itparam, amgparam):
    """fasp_solver_dcsr_krylov_amg (SolCSR.c:476).  x is updated in place (numpy f64).
    Returns the iteration count or a negative ERROR_* code."""
    A, _keep = T.as_csr(ia, ja, a)
    bv, _b = T.as_vec(b)
    assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]
    xv = T.dvector(x.shape[0], x.ctypes.data_as(T.c_double_p))
    return lib().fasp_solver_dcsr_krylov_amg(C.byref(A), C.byref(bv), C.byref(xv),
                                             C.byref(itparam), C.byref(amgparam))


class AMG:
    """Resident hierarchy (fasp_hip_amg).  host_only=True skips the GPU upload."""

    def __init__(self, ia, ja, a, amgparam, host_only=False):
        self._A, self._keep = T.as_csr(ia, ja, a)
        self.h = C.c_void_p()
        fn = lib().fasp_hip_amg_create_host if host_only else lib().fasp_hip_amg_create
        self.status = fn(C.byref(self.h), C.byref(self._A), C.byref(amgparam))
        if self.status < 0:
            self.h = C.c_void_p()
            raise RuntimeError(f"fasp_hip_amg_create failed with status {self.status}")
        self.n = self._A.row

    @classmethod
    def attach(cls, name):
        """Host-only handle on a hierarchy another process of this node published (fasp_hip_amg_attach)."""
        self = cls.__new__(cls)
        self._A = None; self._keep = None
        self.h = C.c_void_p()
        self.status = lib().fasp_hip_amg_attach(C.byref(self.h), name.encode())
        if self.status < 0:
            self.h = C.c_void_p()
            raise RuntimeError(f"fasp_hip_amg_attach({name}) failed with status {self.status}")
        v = T.dCSRmat()
        lib().fasp_hip_amg_get_matrix(self.h, 0, 0, C.byref(v))
        self.n = v.row
        return self

    def publish(self, name):
        st = lib().fasp_hip_amg_publish(self.h, name.encode())
        if st < 0:
            raise RuntimeError(f"fasp_hip_amg_publish({name}) failed with status {st}")

    @staticmethod
    def unpublish(name):
        lib().fasp_hip_amg_unpublish(name.encode())

    @property
    def num_levels(self):
        return lib().fasp_hip_amg_num_levels(self.h)

    def upload(self):
        st = lib().fasp_hip_amg_upload(self.h)
        if st < 0:
            raise RuntimeError(f"fasp_hip_amg_upload failed with status {st}")

    def matrix(self, level, which):
        """which: 0 A, 1 P, 2 R -> (nrow, ncol, ia, ja, val) copies."""
        v = T.dCSRmat()
        st = lib().fasp_hip_amg_get_matrix(self.h, level, which, C.byref(v))
        if st < 0:
            raise IndexError((level, which))
        ia, ja, val = T.csr_arrays(v)
        return v.row, v.col, ia, ja, val

    def cfmark(self, level):
        v = T.ivector()
        if lib().fasp_hip_amg_get_cfmark(self.h, level, C.byref(v)) < 0:
            raise IndexError(level)
        return np.ctypeslib.as_array(v.val, (v.row,)).copy()

    def solve(self, b, itparam, x0=None, hist_cap=600):
        """PCG on the resident hierarchy -> (status, x, hist, stats)."""
        x = np.zeros(self.n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        bv, _b = T.as_vec(b)
        xv, x = T.as_vec(x)
        hist = np.zeros(hist_cap)
        stats = T.fasp_hip_stats()
        st = lib().fasp_hip_solve(self.h, C.byref(bv), C.byref(xv), C.byref(itparam),
                                  T.dp(hist), hist_cap, C.byref(stats))
        return st, x, hist[:max(stats.nhist, 0)].copy(), stats

    def amg_solve(self, b, amgparam=None, x0=None, hist_cap=600):
        """Multigrid cycles as a stand-alone solver (fasp_amg_solve) -> (status, x, hist, stats)."""
        x = np.zeros(self.n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        bv, _b = T.as_vec(b)
        xv, x = T.as_vec(x)
        hist = np.zeros(hist_cap)
        stats = T.fasp_hip_stats()
        st = lib().fasp_hip_amg_solve(self.h, C.byref(bv), C.byref(xv),
                                      C.byref(amgparam) if amgparam is not None else None,
                                      T.dp(hist), hist_cap, C.byref(stats))
        return st, x, hist[:max(min(stats.nhist, hist_cap), 0)].copy(), stats

    def set_rhs(self, b):
        bv, _b = T.as_vec(b)
        st = lib().fasp_hip_set_rhs(self.h, C.byref(bv))
        if st < 0:
            raise RuntimeError(f"fasp_hip_set_rhs failed: {st}")

    def solve_resident(self, itparam, hist_cap=600):
        """Zero initial guess, b and x stay in HBM -> (status, hist, stats)."""
        st = lib().fasp_hip_set_guess(self.h, None)
        if st < 0:
            raise RuntimeError(f"fasp_hip_set_guess failed: {st}")
        hist = np.zeros(hist_cap)
        stats = T.fasp_hip_stats()
        st = lib().fasp_hip_solve_resident(self.h, C.byref(itparam), T.dp(hist), hist_cap,
                                           C.byref(stats))
        return st, hist[:max(stats.nhist, 0)].copy(), stats

    def get_solution(self):
        x = np.zeros(self.n)
        xv, x = T.as_vec(x)
        st = lib().fasp_hip_get_solution(self.h, C.byref(xv))
        if st < 0:
            raise RuntimeError(f"fasp_hip_get_solution failed: {st}")
        return x

    def kernel_info(self, level, which=0):
        """(kernel family, matrix bytes read per pass) of operator which (0 A, 1 P, 2 R) on a level."""
        k = C.c_int(0); b = C.c_double(0)
        if lib().fasp_hip_amg_kernel_info(self.h, level, which, C.byref(k), C.byref(b)) < 0:
            raise IndexError((level, which))
        return k.value, b.value

    def precond(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        st = lib().fasp_hip_precond_amg(self.h, T.dp(r), T.dp(z))
        if st < 0:
            raise RuntimeError(f"fasp_hip_precond_amg failed: {st}")
        return z

    # --- row partition (host side) ---
    def dist_plan(self, rank, nranks, min_rows):
        st = lib().fasp_hip_dist_plan(self.h, rank, nranks, min_rows)
        if st < 0:
            raise RuntimeError(f"fasp_hip_dist_plan failed: {st}")

    def dist_info(self, level):
        info = (C.c_int * 8)()
        if lib().fasp_hip_dist_level_info(self.h, level, info) < 0:
            raise IndexError(level)
        keys = ("replicated", "nglobal", "row0", "nloc", "nghost", "nsend", "first_replicated", "nranks")
        return dict(zip(keys, list(info)))

    def dist_matrix(self, level, which):
        v = T.dCSRmat()
        if lib().fasp_hip_dist_get_matrix(self.h, level, which, C.byref(v)) < 0:
            raise IndexError((level, which))
        ia, ja, val = T.csr_arrays(v)
        return v.row, v.col, ia, ja, val

    def dist_list(self, level, which):
        v = T.ivector()
        if lib().fasp_hip_dist_get_list(self.h, level, which, C.byref(v)) < 0:
            raise IndexError((level, which))
        if v.row == 0:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(v.val, (v.row,)).copy()

    def time_kernel(self, kind, level=0, reps=20):
        return lib().fasp_hip_time_kernel(self.h, kind, level, reps)

    def close(self):
        if self.h:
            lib().fasp_hip_amg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BSRAMG:
    """Resident block hierarchy (fasp_hip_amg_bsr, config 3).  host_only=True skips the GPU upload."""

    def __init__(self, ia, ja, val, nb, amgparam, host_only=False):
        self._A, self._keep = T.as_bsr(ia, ja, val, nb)
        self.h = C.c_void_p()
        fn = lib().fasp_hip_bsr_amg_create_host if host_only else lib().fasp_hip_bsr_amg_create
        self.status = fn(C.byref(self.h), C.byref(self._A), C.byref(amgparam))
        if self.status < 0:
            self.h = C.c_void_p()
            raise RuntimeError(f"fasp_hip_bsr_amg_create failed with status {self.status}")
        self.nb = nb
        self.n = self._A.ROW * nb

    @property
    def num_levels(self):
        return lib().fasp_hip_bsr_amg_num_levels(self.h)

    def matrix(self, level, which):
        """which: 0 A, 1 P, 2 R -> (ROW, COL, NNZ, ia, ja, val) copies."""
        v = T.dBSRmat()
        if lib().fasp_hip_bsr_amg_get_matrix(self.h, level, which, C.byref(v)) < 0:
            raise IndexError((level, which))
        ia = np.ctypeslib.as_array(v.IA, (v.ROW + 1,)).copy()
        ja = np.ctypeslib.as_array(v.JA, (max(v.NNZ, 1),))[:v.NNZ].copy()
        nv = v.NNZ * v.nb * v.nb
        val = np.ctypeslib.as_array(v.val, (max(nv, 1),))[:nv].copy()
        return v.ROW, v.COL, v.NNZ, ia, ja, val

    def diaginv(self, level):
        p = lib().fasp_hip_bsr_amg_get_diaginv(self.h, level)
        if not p:
            return None
        v = T.dBSRmat()
        lib().fasp_hip_bsr_amg_get_matrix(self.h, level, 0, C.byref(v))
        return np.ctypeslib.as_array(p, (v.ROW * v.nb * v.nb,)).copy()

    def solve(self, b, itparam, x0=None, hist_cap=1200):
        """Krylov solve on the resident block hierarchy -> (status, x, hist, stats)."""
        x = np.zeros(self.n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        bv, _b = T.as_vec(b)
        xv, x = T.as_vec(x)
        hist = np.zeros(hist_cap)
        stats = T.fasp_hip_stats()
        st = lib().fasp_hip_bsr_solve(self.h, C.byref(bv), C.byref(xv), C.byref(itparam), T.dp(hist),
                                      hist_cap, C.byref(stats))
        return st, x, hist[:max(min(stats.nhist, hist_cap), 0)].copy(), stats

    def dist_info(self):
        """Row partition of level 0 (one process per GPU): owned block rows [row0, row0 + nloc)."""
        info = (C.c_int * 6)()
        lib().fasp_hip_bsr_dist_info(self.h, info)
        return {"replicated": info[0], "row0": info[1], "nloc": info[2], "nghost": info[3], "first_replicated": info[4], "nb": info[5]}

    def free(self):
        if self.h:
            lib().fasp_hip_bsr_amg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def solver_dbsr_krylov_amg(ia, ja, val, nb, b, x, itparam, amgparam):
    """Drop-in call of fasp_solver_dbsr_krylov_amg (SolBSR.c:349): x is the guess on entry, solution on exit."""
    A, _keep = T.as_bsr(ia, ja, val, nb)
    bv, _b = T.as_vec(b)
    assert x.dtype == np.float64 and x.flags.c_contiguous
    xv = T.dvector(len(x), T.dp(x))
    return lib().fasp_solver_dbsr_krylov_amg(C.byref(A), C.byref(bv), C.byref(xv), C.byref(itparam),
                                             C.byref(amgparam))


def solver_amg(ia, ja, a, b, x, amgparam):
    """Drop-in call of fasp_solver_amg (SolAMG.c:49): x is the guess on entry, solution on exit."""
    A, _keep = T.as_csr(ia, ja, a)
    bv, _b = T.as_vec(b)
    assert x.dtype == np.float64 and x.flags.c_contiguous
    xv = T.dvector(len(x), T.dp(x))
    return lib().fasp_solver_amg(C.byref(A), C.byref(bv), C.byref(xv), C.byref(amgparam))
