"""The reference's own preconditioner objects (base/include/fasp.h:804-981,
base/src/PreCSR.c:46 / :416, base/src/PreDataInit.c:64 / :101): layout of AMG_data / precond_data against the
compiled reference (tests/golden/abi_precond.npz from oracle/ref_shim.c), and the tutorial's caller flow
(tutorial/main/poisson-pcg.c) compiled as plain C against include/fasp_hip.h and run through the C-ABI."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA, ROOT, default_params, orc_solve, read_csr, read_vec

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

AMG_FIELDS = ["max_levels", "num_levels", "A", "R", "P", "b", "x", "Numeric", "pdata", "cfmark", "ILU_levels", "LU",
              "near_kernel_dim", "near_kernel_basis", "SWZ_levels", "Schwarz", "w", "mumps", "cycle_type", "ic", "icmap",
              "colors", "weight"]
PC_FIELDS = ["AMG_type", "print_level", "maxit", "max_levels", "tol", "cycle_type", "smoother", "smooth_order",
             "presmooth_iter", "postsmooth_iter", "relaxation", "polynomial_degree", "coarsening_type", "coarse_solver",
             "coarse_scaling", "amli_degree", "nl_amli_krylov_type", "tentative_smooth", "amli_coef", "mgl_data", "LU", "A",
             "A_nk", "P_nk", "R_nk", "r", "w"]


def test_amg_data_and_precond_data_layout_matches_reference(tmp_path):
    """Every field offset and the sizes, from a C program compiled against include/fasp_hip.h only."""
    z = np.load(os.path.join(G, "abi_precond.npz"))
    src = tmp_path / "off.c"
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "fasp_hip.h"', "int main(void){"]
    for f in AMG_FIELDS:
        lines.append(f'printf("%zu\\n", offsetof(AMG_data, {f}));')
    lines += ['printf("%zu\\n", sizeof(AMG_data));', 'printf("%zu\\n", sizeof(ILU_data));',
              'printf("%zu\\n", sizeof(SWZ_data));', 'printf("%zu\\n", sizeof(ILU_param));']
    for f in PC_FIELDS:
        lines.append(f'printf("%zu\\n", offsetof(precond_data, {f}));')
    lines += ['printf("%zu\\n", sizeof(precond_data));', "return 0;}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "off"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    ref = z["amgdata"].tolist() + z["precdata"].tolist()
    assert out == ref
    assert z["amgdata"][23] == 1104 and z["precdata"][27] == 152


def test_precond_symbols_exported():
    L = fa.lib()
    for s in ("fasp_precond_setup", "fasp_precond_amg", "fasp_precond_famg", "fasp_precond_amli", "fasp_precond_namli",
              "fasp_amg_data_create", "fasp_amg_data_free", "fasp_param_amg_to_prec", "fasp_param_prec_to_amg",
              "fasp_mem_free", "fasp_mem_calloc", "fasp_dvec_alloc", "fasp_dvec_set", "fasp_dvec_free", "fasp_dvec_create",
              "fasp_dcsr_create", "fasp_dcsr_free", "fasp_smoother_dcsr_gs", "fasp_smoother_dcsr_sor",
              "fasp_smoother_dcsr_L1diag"):
        assert hasattr(L, s), s


def test_amg_data_create_free_without_a_device():
    """PreDataInit.c:64: max_levels entries, every one carrying max_levels; a caller-owned array is freed field by field."""
    L = fa.lib()
    L.fasp_amg_data_create.restype = C.c_void_p
    L.fasp_amg_data_create.argtypes = [C.c_short]
    L.fasp_amg_data_free.argtypes = [C.c_void_p, C.c_void_p]
    L.fasp_amg_data_free.restype = None
    p = L.fasp_amg_data_create(5)
    raw = (C.c_char * (5 * 1104)).from_address(p)
    for l in range(5):
        ml, nl = np.frombuffer(raw, np.int16, 2, l * 1104)
        assert ml == 5 and nl == 0
    L.fasp_amg_data_free(p, None)


@pytest.mark.gpu
def test_tutorial_caller_flow_compiled_c(gpu, tmp_path):
    """examples/poisson_pcg.c = the call sequence of tutorial/main/poisson-pcg.c; its iteration table is the
    reference's shipped log tutorial/out/poisson-pcg-c.out (4 iterations, residuals to the printed digits)."""
    exe = tmp_path / "poisson_pcg"
    lib = os.path.join(ROOT, "faspsolver_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "poisson_pcg.c"),
                    "-o", str(exe), "-L", lib, "-lfasp_hip", f"-Wl,-rpath,{lib}"], check=True)
    r = subprocess.run([str(exe), DATA + "/csrmat_FE.dat", DATA + "/rhs_FE.dat"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    # hierarchy seen through mgl_data[l].A: the reference's level sizes (poisson-pcg-c.out)
    lv = re.findall(r"mgl\[(\d)\]: A (\d+) x \d+, (\d+) nonzeros", out)
    assert [(int(a), int(b)) for _, a, b in lv] == [(3969, 27281), (1985, 28523), (541, 7951), (141, 1803)]
    rows = re.findall(r"^\s+(\d+) \|\s+(\S+)\s+\|\s+(\S+)\s+\|", out, re.M)
    assert [(int(i), rr, ar) for i, rr, ar in rows] == [
        (0, "1.000000e+00", "7.514358e+00"), (1, "1.156153e-02", "8.687750e-02"), (2, "3.127181e-04", "2.349876e-03"),
        (3, "4.813471e-06", "3.617014e-05"), (4, "5.312526e-08", "3.992022e-07")]
    assert "status = 4" in out
    # the final line at this revision prints 10 digits (KryUtil.inl:102); the shipped log is older (6 digits)
    m = re.search(r"Number of iterations = 4 with relative residual (\S+)\.\n", out)
    assert m and "%.6e" % float(m.group(1)) == "5.312526e-08"


@pytest.mark.gpu
def test_precond_setup_objects_through_ctypes(gpu):
    """fasp_precond_setup(PREC_AMG) -> precond_data / AMG_data views equal the handle API's hierarchy; fasp_precond_amg
    through the function pointer equals fasp_hip_precond_amg; pcdata fields are re-read at every application."""
    L = fa.lib()
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    itp, amgp = default_params(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667; amgp.print_level = 0
    A, keep = T.as_csr(ia, ja, a)
    L.fasp_precond_setup.restype = C.POINTER(T.precond)
    L.fasp_precond_setup.argtypes = [C.c_short, C.POINTER(T.AMG_param), C.c_void_p, C.POINTER(T.dCSRmat)]
    pc = L.fasp_precond_setup(2, C.byref(amgp), None, C.byref(A))
    assert pc and pc.contents.fct
    raw = (C.c_char * 152).from_address(pc.contents.data)
    mgl_ptr = int(np.frombuffer(raw, np.uint64, 1, 80)[0])
    maxit_off = 4
    nl = int(np.frombuffer((C.c_char * 4).from_address(mgl_ptr), np.int16, 2)[1])
    H = fa.AMG(ia, ja, a, amgp)
    assert nl == H.num_levels
    for l in range(nl):
        v = T.dCSRmat.from_address(mgl_ptr + l * 1104 + 8)
        r_, c_, ia2, ja2, a2 = H.matrix(l, 0)
        i3, j3, a3 = T.csr_arrays(v)
        assert (v.row, v.col, v.nnz) == (r_, c_, len(a2))
        assert np.array_equal(i3, ia2) and np.array_equal(j3, ja2) and np.array_equal(a3, a2)
    r = np.random.default_rng(3).standard_normal(len(f))
    z = np.zeros_like(r)
    fct = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)(C.cast(pc.contents.fct, C.c_void_p).value)
    fct(T.dp(r), T.dp(z), pc.contents.data)
    assert np.array_equal(z, H.precond(r))
    # two cycles per application: maxit lives in precond_data and is read at every call (PreCSR.c:421)
    np.frombuffer(raw, np.int32, 1, maxit_off)[0] = 2
    z2 = np.zeros_like(r)
    fct(T.dp(r), T.dp(z2), pc.contents.data)
    assert not np.array_equal(z2, z) and np.linalg.norm(z2 - z) > 1e-8 * np.linalg.norm(z)
    np.frombuffer(raw, np.int32, 1, maxit_off)[0] = 1
    # the reference's Krylov entry point with this object: equal to the oracle's solve
    x = np.zeros(len(f)); bv, _f = T.as_vec(f); xv, x = T.as_vec(x)
    st = L.fasp_solver_dcsr_pcg(C.byref(A), C.byref(bv), C.byref(xv), pc, 1e-8, 1e-18, 100, 1, 0)
    itp.tol = 1e-8; itp.maxit = 100
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert st == s1 and np.abs(x - x1).max() <= 1e-9 * np.abs(x1).max()
    L.fasp_amg_data_free.argtypes = [C.c_void_p, C.c_void_p]; L.fasp_amg_data_free.restype = None
    L.fasp_mem_free.argtypes = [C.c_void_p]; L.fasp_mem_free.restype = None
    L.fasp_amg_data_free(mgl_ptr, C.byref(amgp))
    L.fasp_mem_free(pc.contents.data)
    L.fasp_mem_free(C.cast(pc, C.c_void_p))
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["gs_fwd", "gs_bwd", "sor_fwd", "sor_bwd", "l1"])
def test_standalone_sequential_smoothers(gpu, which):
    """fasp_smoother_dcsr_gs / _sor / _L1diag as exported sweeps equal the sequential host sweeps
    (ItrSmootherCSR.c:327-334, :981-993, :1560-1574)."""
    L = fa.lib()
    ia, ja, a, f, ue = fa.poisson7pt(9)
    n = len(f)
    rng = np.random.default_rng(17)
    u0 = rng.standard_normal(n)
    A, keep = T.as_csr(ia, ja, a)
    u = u0.copy(); uv, u = T.as_vec(u); bv, _f = T.as_vec(f)
    w = 1.1
    s = -1 if which.endswith("bwd") else 1
    i1, i2 = (0, n - 1) if s > 0 else (n - 1, 0)
    ref = u0.copy()
    order = range(n) if s > 0 else range(n - 1, -1, -1)
    for sweep in range(2):
        if which == "l1":
            new = ref.copy()
            for i in range(n):
                t = f[i]; d = 0.0
                for k in range(ia[i], ia[i + 1]):
                    t -= a[k] * ref[ja[k]]; d += abs(a[k])
                new[i] = ref[i] + t / d
            ref = new
        else:
            for i in order:
                t = f[i]; d = 0.0
                for k in range(ia[i], ia[i + 1]):
                    if ja[k] != i: t -= a[k] * ref[ja[k]]
                    else: d = a[k]
                ref[i] = t * (1.0 / d) if which.startswith("gs") else w * (t / d) + (1 - w) * ref[i]
    if which.startswith("gs"):
        L.fasp_smoother_dcsr_gs.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int]
        L.fasp_smoother_dcsr_gs.restype = None
        L.fasp_smoother_dcsr_gs(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2)
    elif which.startswith("sor"):
        L.fasp_smoother_dcsr_sor.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int, C.c_double]
        L.fasp_smoother_dcsr_sor.restype = None
        L.fasp_smoother_dcsr_sor(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2, w)
    else:
        L.fasp_smoother_dcsr_L1diag.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int]
        L.fasp_smoother_dcsr_L1diag.restype = None
        L.fasp_smoother_dcsr_L1diag(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2)
    assert np.abs(u - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["gs_fwd", "gs_bwd", "sor_fwd", "sor_bwd"])
def test_standalone_sweeps_on_a_ragged_matrix(gpu, which):
    """The split sweep (a parallel pass + a sparse triangular solve, csrc/seq_split.hip.h) on what a stencil never shows:
    an unsymmetric pattern (anti-dependencies), rows far longer than the slot storage holds (tails), empty rows, rows
    without a diagonal and with a zero diagonal (left alone, ItrSmootherCSR.c: |d| <= SMALLREAL), a diagonal stored
    twice (the reference's loop keeps the LAST one and subtracts nothing for either)."""
    import scipy.sparse as sp
    L = fa.lib()
    rng = np.random.default_rng(5)
    n = 700
    M = sp.random(n, n, density=0.03, random_state=7, format="lil")
    for i in (3, 250, 699):                       # very long rows: more lower entries than 8 x 64 slots
        M[i, :] = rng.standard_normal(n)
    for i in range(n): M[i, i] = 30.0 + rng.random()
    for i in (10, 11, 400): M[i, :] = 0.0         # empty rows
    M[20, 20] = 0.0                               # stored zero diagonal
    M = M.tocsr(); M.sort_indices()
    M[21, 21] = 0.0; M.eliminate_zeros()          # no diagonal at all in row 21 (row 20 lost its zero too: put it back below)
    ia = M.indptr.astype(np.int32).tolist(); ja = M.indices.astype(np.int32).tolist(); a = M.data.tolist()
    def insert(row, col, v):                      # append an entry to a row (unsorted storage is allowed)
        k = ia[row + 1]
        ja.insert(k, col); a.insert(k, v)
        for r in range(row + 1, n + 1): ia[r] += 1
    insert(20, 20, 0.0)                           # zero diagonal, stored
    insert(100, 100, 55.0)                        # diagonal stored twice: the last one counts
    ia = np.array(ia, dtype=np.int32); ja = np.array(ja, dtype=np.int32); a = np.array(a)
    f = rng.standard_normal(n); u0 = rng.standard_normal(n)
    A, keep = T.as_csr(ia, ja, a)
    u = u0.copy(); uv, u = T.as_vec(u); bv, _f = T.as_vec(f)
    w = 1.1
    s = -1 if which.endswith("bwd") else 1
    i1, i2 = (0, n - 1) if s > 0 else (n - 1, 0)
    ref = u0.copy()
    order = range(n) if s > 0 else range(n - 1, -1, -1)
    for sweep in range(2):
        for i in order:
            t = f[i]; d = 0.0
            for k in range(ia[i], ia[i + 1]):
                if ja[k] != i: t -= a[k] * ref[ja[k]]
                else: d = a[k]
            if abs(d) > 1e-20:
                ref[i] = t * (1.0 / d) if which.startswith("gs") else w * (t / d) + (1 - w) * ref[i]
    if which.startswith("gs"):
        L.fasp_smoother_dcsr_gs.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int]
        L.fasp_smoother_dcsr_gs.restype = None
        L.fasp_smoother_dcsr_gs(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2)
    else:
        L.fasp_smoother_dcsr_sor.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int, C.c_double]
        L.fasp_smoother_dcsr_sor.restype = None
        L.fasp_smoother_dcsr_sor(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2, w)
    assert np.all(np.isfinite(u))
    assert np.abs(u - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["gs_fwd", "sor_bwd"])
def test_standalone_sweeps_with_distant_couplings(gpu, which):
    """A 5-point grid in natural order (anti-diagonal dependency classes, every row reads the class before) plus twenty
    couplings 19 000 rows back, with the strips of the dataflow triangular solve cut to 16 KB
    (fasp_hip_tune("seq_strip_kb")): a few dozen strips, every one reading its predecessor's last grid line and twenty of
    them a row far behind -- ghosts that travel through W in memory (FASP_HIP_SETUP_TIMING=1 prints the counts)."""
    import scipy.sparse as sp
    L = fa.lib()
    nx, ny = 200, 100
    n = nx * ny
    rng = np.random.default_rng(9)
    I = sp.identity(nx, format="csr"); J = sp.identity(ny, format="csr")
    Tx = sp.diags([-1.0, -1.0], [-1, 1], shape=(nx, nx)); Ty = sp.diags([-1.0, -1.0], [-1, 1], shape=(ny, ny))
    M = (sp.kron(J, Tx) + sp.kron(Ty, I) + 4.5 * sp.identity(n)).tolil()
    for i in rng.choice(np.arange(19500, n), size=20, replace=False):
        M[i, i - 19000] = -0.25; M[i - 19000, i] = -0.25
    M = M.tocsr(); M.sort_indices()
    ia = M.indptr.astype(np.int32); ja = M.indices.astype(np.int32); a = M.data.copy()
    f = rng.standard_normal(n); u0 = rng.standard_normal(n)
    A, keep = T.as_csr(ia, ja, a)
    u = u0.copy(); uv, u = T.as_vec(u); bv, _f = T.as_vec(f)
    w = 1.1
    s = -1 if which.endswith("bwd") else 1
    i1, i2 = (0, n - 1) if s > 0 else (n - 1, 0)
    ref = u0.copy()
    order = range(n) if s > 0 else range(n - 1, -1, -1)
    ial, jal, al = ia.tolist(), ja.tolist(), a.tolist()
    for sweep in range(2):
        for i in order:
            t = f[i]; d = 0.0
            for k in range(ial[i], ial[i + 1]):
                if jal[k] != i: t -= al[k] * ref[jal[k]]
                else: d = al[k]
            ref[i] = t * (1.0 / d) if which.startswith("gs") else w * (t / d) + (1 - w) * ref[i]
    try:
        L.fasp_hip_tune(b"seq_strip_kb", 16)
        if which.startswith("gs"):
            L.fasp_smoother_dcsr_gs.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int]
            L.fasp_smoother_dcsr_gs.restype = None
            L.fasp_smoother_dcsr_gs(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2)
        else:
            L.fasp_smoother_dcsr_sor.argtypes = [C.POINTER(T.dvector), C.c_int, C.c_int, C.c_int, C.POINTER(T.dCSRmat), C.POINTER(T.dvector), C.c_int, C.c_double]
            L.fasp_smoother_dcsr_sor.restype = None
            L.fasp_smoother_dcsr_sor(C.byref(uv), i1, i2, s, C.byref(A), C.byref(bv), 2, w)
    finally:
        L.fasp_hip_tune(b"seq_strip_kb", 0)
    assert np.abs(u - ref).max() <= 1e-12 * np.abs(ref).max()
