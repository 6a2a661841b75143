"""ini front-end (fasp_hip_param_input = fasp_param_input + fasp_param_init, AuxInput.c:86 / AuxParam.c:34)
against the reference on the reference's own ini files (data fixtures under tests/golden/data/ini), and the
Fortran-style wrapper (SolWrapper.c:261)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA, ROOT, default_params, have_ref, orc_solve, poisson7pt, ref

INI = sorted(glob.glob(os.path.join(DATA, "ini", "*.dat")))
G = os.path.join(ROOT, "tests", "golden")


def parse(fname):
    itp, amgp = T.ITS_param(), T.AMG_param()
    st = fa.lib().fasp_hip_param_input(fname.encode() if fname else None, C.byref(itp), C.byref(amgp))
    return st, itp, amgp


def masked(amgp):
    """AMG_param bytes without polynomial_degree (unset by the reference's input defaults) and padding."""
    b = bytearray(bytes(amgp))
    o = T.AMG_param.polynomial_degree.offset
    b[o:o + 2] = b"\0\0"
    return bytes(b)


FIELDS_AMG = [f for f, _ in T.AMG_param._fields_ if f not in ("polynomial_degree", "amli_coef")]
FIELDS_ITS = [f for f, _ in T.ITS_param._fields_]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("fname", INI, ids=[os.path.basename(f) for f in INI])
def test_parser_matches_reference(fname):
    R = ref()
    R.ref_param_from_file.argtypes = [C.c_char_p, C.POINTER(T.ITS_param), C.POINTER(T.AMG_param)]
    i2, a2 = T.ITS_param(), T.AMG_param()
    R.ref_param_from_file(fname.encode(), C.byref(i2), C.byref(a2))
    st, i1, a1 = parse(fname)
    assert st == 0
    for f in FIELDS_ITS:
        assert getattr(i1, f) == getattr(i2, f), f
    for f in FIELDS_AMG:
        assert getattr(a1, f) == getattr(a2, f), f


@pytest.mark.parametrize("fname", INI, ids=[os.path.basename(f) for f in INI])
def test_parser_matches_golden(fname):
    z = np.load(os.path.join(G, "ini_params.npz"))
    key = os.path.basename(fname)
    st, i1, a1 = parse(fname)
    assert st == 0
    assert bytes(i1) == z[key + "_its"].tobytes()
    assert masked(a1) == z[key + "_amg"].tobytes()


def test_defaults_errors_and_unknown_keys(tmp_path):
    st, itp, amgp = parse(None)
    assert st == 0 and itp.itsolver_type == T.SOLVER_CG and itp.restart == 25 and itp.maxit == 500
    assert amgp.smoother == T.SMOOTHER_GS and amgp.aggregation_type == 1 and amgp.strong_coupled == 0.25
    assert parse(str(tmp_path / "missing.dat"))[0] == -10           # ERROR_OPEN_FILE
    bad = tmp_path / "bad.dat"
    bad.write_text("AMG_type = XX\n")
    assert parse(str(bad))[0] == T.ERROR_INPUT_PAR
    bad.write_text("stop_type = 7\n")
    assert parse(str(bad))[0] == T.ERROR_INPUT_PAR                   # fasp_param_check
    ok = tmp_path / "ok.dat"
    ok.write_text("% comment\n[section]\nno_such_key = 3\nAMG_smoother = JACOBI % trailing text\n"
                  "AMG_relaxation = 0.6667\nAMG_cycle_type = w\nAMG_coarse_scaling = On\nsolver_type = 6\n")
    st, itp, amgp = parse(str(ok))
    assert st == 0 and amgp.smoother == T.SMOOTHER_JACOBI and amgp.relaxation == 0.6667
    assert amgp.cycle_type == T.W_CYCLE and amgp.coarse_scaling == 1 and itp.itsolver_type == 6


@pytest.mark.gpu
def test_fortran_wrapper(tmp_path, monkeypatch):
    """CALL FASP_FWRAPPER_DCSR_KRYLOV_AMG: parameters from ini/amg.dat of the working directory."""
    (tmp_path / "ini").mkdir()
    (tmp_path / "ini" / "amg.dat").write_text(
        "solver_type = 1\nprecond_type = 2\nAMG_type = C\nAMG_smoother = JACOBI\nAMG_relaxation = 0.6667\n"
        "AMG_cycle_type = V\nstop_type = 1\n")
    monkeypatch.chdir(tmp_path)
    ia, ja, a, f, ue = poisson7pt(16)
    itp, amgp = default_params(); itp.tol = 1e-8; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, itp, amgp)
    n = C.c_int(len(f)); nnz = C.c_int(len(a)); tol = C.c_double(1e-8); maxit = C.c_int(500); prt = C.c_int(0)
    u = np.zeros(len(f)); ia2 = ia.copy(); ja2 = ja.copy(); a2 = a.copy(); f2 = f.copy()
    fa.lib().fasp_fwrapper_dcsr_krylov_amg_(C.byref(n), C.byref(nnz), ia2.ctypes.data_as(T.c_int_p),
                                            ja2.ctypes.data_as(T.c_int_p), T.dp(a2), T.dp(f2), T.dp(u),
                                            C.byref(tol), C.byref(maxit), C.byref(prt))
    assert np.abs(u - x1).max() <= 1e-10 * np.abs(x1).max()


def test_file_readers_on_shipped_data(tmp_path):
    """fasp_dcsrvec_read2 / fasp_dvec_read / fasp_dbsr_read (BlaIO.c:164/:938/:807) on the reference's data files."""
    from _libs import read_csr, read_vec, read_bsr
    L = fa.lib()
    A = T.dCSRmat(); b = T.dvector()
    assert L.fasp_dcsrvec_read2((DATA + "/csrmat_FE.dat").encode(), (DATA + "/rhs_FE.dat").encode(), C.byref(A), C.byref(b)) == 0
    ia, ja, a = read_csr(DATA + "/csrmat_FE.dat"); f = read_vec(DATA + "/rhs_FE.dat")
    i2, j2, a2 = T.csr_arrays(A)
    assert (A.row, A.col, A.nnz) == (len(ia) - 1, len(ia) - 1, len(a))
    assert np.array_equal(i2, ia) and np.array_equal(j2, ja) and np.array_equal(a2, a)
    assert np.array_equal(np.ctypeslib.as_array(b.val, (b.row,)), f)
    L.fasp_hip_free_system(C.byref(A), C.byref(b), None)
    v = T.dvector()
    assert L.fasp_dvec_read((DATA + "/rhs_SPE01.dat").encode(), C.byref(v)) == 0
    assert np.array_equal(np.ctypeslib.as_array(v.val, (v.row,)), read_vec(DATA + "/rhs_SPE01.dat"))
    B = T.dBSRmat()
    assert L.fasp_dbsr_read((DATA + "/bsrmat_SPE01.dat").encode(), C.byref(B)) == 0
    bi, bj, bv, nb = read_bsr(DATA + "/bsrmat_SPE01.dat")
    assert (B.ROW, B.NNZ, B.nb) == (len(bi) - 1, len(bj), nb)
    assert np.array_equal(np.ctypeslib.as_array(B.IA, (B.ROW + 1,)), bi)
    assert np.array_equal(np.ctypeslib.as_array(B.JA, (B.NNZ,)), bj)
    assert np.array_equal(np.ctypeslib.as_array(B.val, (B.NNZ * nb * nb,)), bv)
    L.fasp_hip_free_bsr(C.byref(B))
    # errors instead of exits
    assert L.fasp_dvec_read(str(tmp_path / "none.dat").encode(), C.byref(v)) == -10
    bad = tmp_path / "bad.dat"; bad.write_text("3\n1 2\n")
    assert L.fasp_dcsrvec_read2(str(bad).encode(), str(bad).encode(), C.byref(A), C.byref(b)) == -11


@pytest.mark.gpu
def test_c_driver_from_files(tmp_path):
    """examples/solve_from_files.c: ini file + data files -> fasp_solver_dcsr_krylov_amg, plain C against the header."""
    import subprocess
    exe = tmp_path / "solve"
    lib = os.path.join(ROOT, "faspsolver_amd")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "solve_from_files.c"),
                    "-o", str(exe), "-L", lib, "-lfasp_hip", f"-Wl,-rpath,{lib}", "-lm"], check=True)
    r = subprocess.run([str(exe), os.path.join(ROOT, "examples", "amg_jacobi.dat"), DATA + "/csrmat_FE.dat",
                        DATA + "/rhs_FE.dat"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("status")][0]
    its = int(line.split("=")[1].split(",")[0]); rel = float(line.split("=")[-1])
    ia, ja, a = __import__("_libs").read_csr(DATA + "/csrmat_FE.dat"); f = __import__("_libs").read_vec(DATA + "/rhs_FE.dat")
    itp, amgp = default_params(); itp.tol = 1e-8; itp.maxit = 100; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    s1, x1, h1, rr = orc_solve(ia, ja, a, f, itp, amgp)
    assert its == s1 and rel < 1e-8


def _csr_of(A):
    i2, j2, a2 = T.csr_arrays(A)
    return (A.row, A.col, A.nnz), i2.copy(), j2.copy(), a2.copy()


def test_coordinate_readers_and_writers(tmp_path):
    """fasp_dcoo_read / _read1 / _shift_read / fasp_dmtx_read / fasp_dmtxsym_read (BlaIO.c:332-697) and the
    writers fasp_dcsr_write_coo / fasp_dvec_write / fasp_dcsrvec_write2 (:1623/:1388/:1145): every row
    keeps the file's order of its entries (fasp_format_dcoo_dcsr is a stable counting sort)."""
    L = fa.lib()
    ia, ja, a, f, ue = poisson7pt(5)
    A, keep = T.as_csr(ia, ja, a)
    p = str(tmp_path / "A.coo").encode()
    assert L.fasp_dcsr_write_coo(p, C.byref(A)) == 0
    # the reference's writer puts the sizes into a COMMENT line, which its readers skip: hand the
    # sizes over as a data line to read the triples back
    lines = open(p).read().splitlines()
    assert lines[0] == "%% dimension of the matrix and nonzeros %d  %d  %d" % (len(f), len(f), len(a))
    assert lines[1] == "1 1 %+.15E" % a[0]
    open(p, "w").write("%d %d %d\n" % (len(f), len(f), len(a)) + "\n".join(lines[1:]) + "\n")
    for fn in (L.fasp_dcoo_read1, L.fasp_dcoo_shift_read, L.fasp_dmtx_read):
        B = T.dCSRmat()
        assert fn(p, C.byref(B)) == 0
        dims, i2, j2, a2 = _csr_of(B)
        assert dims == (len(f), len(f), len(a))
        assert np.array_equal(i2, ia) and np.array_equal(j2, ja) and np.array_equal(a2, a)
        L.fasp_hip_free_system(C.byref(B), None, None)
    # 0-based triples in scrambled order: rows sorted, order inside a row as in the file
    z = tmp_path / "z.coo"
    z.write_text("% comment\n3 3 5\n2 2 5.5\n0 1 -1\n0 0 4\n1 1 3\n2 0 7e-1\n")
    B = T.dCSRmat()
    assert L.fasp_dcoo_read(str(z).encode(), C.byref(B)) == 0
    dims, i2, j2, a2 = _csr_of(B)
    assert dims == (3, 3, 5) and list(i2) == [0, 2, 3, 5] and list(j2) == [1, 0, 1, 2, 0]
    assert list(a2) == [-1.0, 4.0, 3.0, 5.5, 0.7]
    L.fasp_hip_free_system(C.byref(B), None, None)
    # one triangle of a symmetric MatrixMarket file (header counts the stored entries)
    s = tmp_path / "s.mtx"
    s.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 5\n1 1 2\n2 1 -1\n2 2 2\n3 2 -1\n3 3 2\n")
    assert L.fasp_dmtxsym_read(str(s).encode(), C.byref(B)) == 0
    dims, i2, j2, a2 = _csr_of(B)
    assert dims == (3, 3, 7) and list(i2) == [0, 2, 5, 7]
    assert list(j2) == [0, 1, 0, 1, 2, 1, 2] and list(a2) == [2, -1, -1, 2, -1, -1, 2]
    L.fasp_hip_free_system(C.byref(B), None, None)
    # vector and two-file CSR round trips (the reference writes %le for the latter: 7 digits)
    v, vk = T.as_vec(f)
    pv = str(tmp_path / "v.dat").encode()
    assert L.fasp_dvec_write(pv, C.byref(v)) == 0
    w = T.dvector()
    assert L.fasp_dvec_read(pv, C.byref(w)) == 0
    assert np.allclose(np.ctypeslib.as_array(w.val, (w.row,)), f, rtol=1e-15, atol=0)
    pm, pr = str(tmp_path / "m.dat").encode(), str(tmp_path / "r.dat").encode()
    assert L.fasp_dcsrvec_write2(pm, pr, C.byref(A), C.byref(v)) == 0
    A2 = T.dCSRmat(); b2 = T.dvector()
    assert L.fasp_dcsrvec_read2(pm, pr, C.byref(A2), C.byref(b2)) == 0
    dims, i2, j2, a2 = _csr_of(A2)
    assert np.array_equal(i2, ia) and np.array_equal(j2, ja) and np.allclose(a2, a, rtol=1e-6)
    L.fasp_hip_free_system(C.byref(A2), C.byref(b2), None)
    # errors instead of exits
    assert L.fasp_dcoo_read(str(tmp_path / "none").encode(), C.byref(B)) == -10
    bad = tmp_path / "bad.coo"; bad.write_text("2 2 3\n0 0 1\n5 0 1\n")
    assert L.fasp_dcoo_read(str(bad).encode(), C.byref(B)) == -11


def test_coordinate_readers_equal_reference(tmp_path):
    from _libs import have_ref, ref
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    R = ref(); L = fa.lib()
    rng = np.random.default_rng(3)
    m, nnz = 40, 300
    ri = rng.integers(1, m + 1, nnz); ci = rng.integers(1, m + 1, nnz); v = rng.standard_normal(nnz)
    p = tmp_path / "r.mtx"
    p.write_text("%% random\n%d %d %d\n" % (m, m, nnz) + "".join("%d %d %.17e\n" % t for t in zip(ri, ci, v)))
    for name in ("fasp_dcoo_read1", "fasp_dmtx_read", "fasp_dcoo_shift_read"):
        A = T.dCSRmat(); B = T.dCSRmat()
        getattr(R, name).restype = None
        getattr(R, name)(str(p).encode(), C.byref(A))
        assert getattr(L, name)(str(p).encode(), C.byref(B)) == 0
        d1, i1, j1, a1 = _csr_of(A); d2, i2, j2, a2 = _csr_of(B)
        assert d1 == d2 and np.array_equal(i1, i2) and np.array_equal(j1, j2) and np.array_equal(a1, a2)
        L.fasp_hip_free_system(C.byref(B), None, None)


@pytest.mark.gpu
def test_fortran_wrappers_amg_and_bsr():
    """SolWrapper.c:136 (AMG as the solver) and :397 (block matrix; the wrapper's defaults ask for
    pairwise aggregation)."""
    from _libs import oracle
    ia, ja, a, f, ue = poisson7pt(16)
    n = C.c_int(len(f)); nnz = C.c_int(len(a)); tol = C.c_double(1e-8); maxit = C.c_int(100); prt = C.c_int(0)
    u = np.zeros(len(f)); ia2 = ia.copy(); ja2 = ja.copy(); a2 = a.copy(); f2 = f.copy()
    fa.lib().fasp_fwrapper_dcsr_amg_(C.byref(n), C.byref(nnz), ia2.ctypes.data_as(T.c_int_p), ja2.ctypes.data_as(T.c_int_p),
                                     T.dp(a2), T.dp(f2), T.dp(u), C.byref(tol), C.byref(maxit), C.byref(prt))
    amgp = fa.param_amg_init(); amgp.tol = 1e-8; amgp.maxit = 100; amgp.print_level = 0
    x = np.zeros(len(f))
    st = fa.solver_amg(ia, ja, a, f, x, amgp)
    assert st > 0 and np.array_equal(u, x)
