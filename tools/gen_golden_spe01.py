"""Pin SPE01 (config 3's shipped matrix, /root/reference/data/bsrmat_SPE01.dat) BEFORE rounding is amplified.
Written from the REFERENCE ITSELF (oracle/_ref/libfasp_ref.so); run in the build container only:

    python tools/gen_golden_spe01.py     ->  tests/golden/spe01_pin.npz

  spe01_z        ONE application of the block preconditioner on rhs_SPE01, z = B r: fasp_precond_dbsr_amg (PreBSR.c:1149) on the hierarchy
                 of fasp_amg_setup_ua_bsr with config 3's parameters -- SPE01 does not coarsen, so this is the coarse solver of
                 fasp_solver_mgcycle_bsr: fasp_solver_dbsr_pvgmres(A, b, x, NULL, tol, ..., min(n^2, 200), restart 25)
  spe01_inner_x  the iterates x_k, k = 1 .. 25, of that inner GMRES(25) (first restart cycle): fasp_solver_dbsr_pvgmres with MaxIt = k
  spe01_inner_res   ||f - A x_k||_2 / ||f||_2 recomputed with the reference's own fasp_blas_dbsr_aAxpy + norm2
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import DATA, T, bsr_params, bsr_protos, read_bsr, read_vec, ref  # noqa: E402

R = ref()
assert R is not None, "oracle/_ref/libfasp_ref.so missing: run `make -C oracle` with /root/reference present"
bsr_protos()
ia, ja, val, nb = read_bsr(DATA + "/bsrmat_SPE01.dat"); f = read_vec(DATA + "/rhs_SPE01.dat")
A, keep = T.as_bsr(ia, ja, val, nb)
n = A.ROW * nb
out = {}

itp, amgp = bsr_params()
h = R.ref_bsr_setup_ua(C.byref(A), C.byref(amgp))
assert R.ref_bsr_num_levels(h) == 1
R.ref_bsr_precond.argtypes = [C.c_void_p, C.POINTER(T.AMG_param), C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
z = np.zeros(n); r = f.copy()
R.ref_bsr_precond(h, C.byref(amgp), C.byref(A), T.dp(r), T.dp(z))
out["spe01_z"] = z
out["spe01_amg_tol"] = np.array(amgp.tol); out["spe01_amg_maxit"] = np.array(amgp.maxit)

fn = R.fasp_solver_dbsr_pvgmres
fn.argtypes = [C.POINTER(T.dBSRmat), C.POINTER(T.dvector), C.POINTER(T.dvector), C.c_void_p, C.c_double, C.c_double, C.c_int,
               C.c_short, C.c_short, C.c_short]
R.fasp_blas_dbsr_aAxpy.argtypes = [C.c_double, C.POINTER(T.dBSRmat), T.c_double_p, T.c_double_p]
R.fasp_blas_darray_norm2.restype = C.c_double; R.fasp_blas_darray_norm2.argtypes = [C.c_int, T.c_double_p]
nf = R.fasp_blas_darray_norm2(n, T.dp(f))
xs, res = [], []
for k in range(1, 26):
    x = np.zeros(n); bv, fk = T.as_vec(f); xv = T.dvector(n, T.dp(x))
    st = fn(C.byref(A), C.byref(bv), C.byref(xv), None, 1e-30, 1e-300, k, 25, 1, 0)
    rr = f.copy()
    R.fasp_blas_dbsr_aAxpy(-1.0, C.byref(A), T.dp(x), T.dp(rr))
    xs.append(x.copy()); res.append(R.fasp_blas_darray_norm2(n, T.dp(rr)) / nf)
out["spe01_inner_x"] = np.array(xs); out["spe01_inner_res"] = np.array(res)
print("inner residuals:", " ".join(f"{v:.3e}" for v in res))
print("|z|_max", np.abs(z).max())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spe01_pin.npz"), **out)
