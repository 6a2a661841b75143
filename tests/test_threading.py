"""Threading contract of the boundary (SURVEY.md section 8b "Threading"; the serial reference lets different host threads
solve on disjoint data, AuxThreads.c:29-57).  The library serialises its computing entry points behind one recursive
process-wide lock (csrc/solver.hip, FASP_ENTRY): calls from several threads may be slow, never wrong.

CPU part: the host-only entries (ini parser, host AMG setup) called from several threads at once give what they give
one after the other.  GPU part: two threads solving different systems through fasp_solver_dcsr_krylov_amg at the same
time get the iteration counts and solutions of the same solves run alone.
"""
import ctypes as C
import glob
import os
import threading

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

from _libs import DATA

INI = sorted(glob.glob(os.path.join(DATA, "ini", "*.dat")))


def _parse(fname):
    itp, amgp = T.ITS_param(), T.AMG_param()
    st = fa.lib().fasp_hip_param_input(fname.encode(), C.byref(itp), C.byref(amgp))
    return st, bytes(itp), bytes(amgp)


def _run_threads(jobs):
    """jobs: list of callables; each runs in a thread of its own, all released together; returns their results in order."""
    out = [None] * len(jobs)
    err = []
    gate = threading.Barrier(len(jobs))

    def work(k):
        try:
            gate.wait(60)
            out[k] = jobs[k]()
        except Exception as e:  # noqa: BLE001
            err.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    assert not err, err
    assert all(not t.is_alive() for t in th)
    return out


def test_ini_parser_from_many_threads():
    serial = [_parse(f) for f in INI]
    for _ in range(3):
        got = _run_threads([lambda f=f: _parse(f) for f in INI])
        assert got == serial


def _hierarchy_digest(n, smoother):
    ia, ja, a, f, ue = fa.poisson7pt(n)
    amgp = fa.param_amg_init()
    amgp.smoother = smoother
    H = fa.AMG(ia, ja, a, amgp, host_only=True)
    dig = []
    for l in range(H.num_levels):
        for which in (0, 1, 2):
            try:
                r, c, mia, mja, mv = H.matrix(l, which)
            except IndexError:
                continue
            dig.append((l, which, r, c, mia.tobytes(), mja.tobytes(), mv.tobytes()))
    H.close()
    return dig


def test_host_setup_from_two_threads_is_the_serial_hierarchy():
    sizes = (12, 17)
    serial = [_hierarchy_digest(n, T.SMOOTHER_JACOBI) for n in sizes]
    got = _run_threads([lambda n=n: _hierarchy_digest(n, T.SMOOTHER_JACOBI) for n in sizes])
    assert got == serial


def _solve(n, smoother, relax):
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp = fa.param_solver_init(); itp.tol = 1e-8; itp.print_level = 0
    amgp = fa.param_amg_init(); amgp.smoother = smoother; amgp.relaxation = relax
    x = np.zeros(len(f))
    st = fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, itp, amgp)
    return st, x


@pytest.mark.gpu
def test_two_threads_solving_disjoint_systems(gpu):
    """fasp_solver_dcsr_krylov_amg (SolCSR.c:476) from two host threads at once, different matrices and smoothers:
    serialised inside the library, bit-identical to the same solves run one after the other."""
    cases = [(40, T.SMOOTHER_JACOBI, 0.6667), (33, T.SMOOTHER_GS, 1.0), (28, T.SMOOTHER_SOR, 1.1), (36, T.SMOOTHER_JACOBI, 0.8)]
    serial = [_solve(*c) for c in cases]
    for _ in range(2):
        got = _run_threads([lambda c=c: _solve(*c) for c in cases])
        for (st, x), (st0, x0) in zip(got, serial):
            assert st == st0 and st > 0
            assert np.array_equal(x, x0)


@pytest.mark.gpu
def test_blas_entries_from_two_threads(gpu):
    """fasp_blas_dcsr_mxv / fasp_blas_darray_dotprod (BlaSpmvCSR.c:242, BlaArray.c:771) hammered from two threads: every call
    returns what it returns alone (the context's reduction buffers are never shared between two calls in flight)."""
    L = fa.lib()
    rng = np.random.default_rng(7)
    work = []
    for n in (30, 37):
        ia, ja, a, f, ue = fa.poisson7pt(n)
        x = rng.standard_normal(len(f))
        A, keep = T.as_csr(ia, ja, a)
        work.append((A, keep, x, len(f)))

    def run(k):
        A, keep, x, m = work[k]
        y = np.zeros(m)
        res = []
        for _ in range(20):
            L.fasp_blas_dcsr_mxv(C.byref(A), T.dp(x), T.dp(y))
            res.append((y.copy(), L.fasp_blas_darray_dotprod(m, T.dp(x), T.dp(y))))
        return res

    serial = [run(0), run(1)]
    got = _run_threads([lambda: run(0), lambda: run(1)])
    for g, s in zip(got, serial):
        for (y, d), (y0, d0) in zip(g, s):
            assert np.array_equal(y, y0) and d == d0
