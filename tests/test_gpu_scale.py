"""Parity at scale (fixture F5): the HIP path through the C-ABI at 64^3, 128^3 and
256^3 against numbers the REFERENCE produced -- tests/golden/p7_scale.npz, written by
tools/gen_golden_f5.py from the compiled reference (64^3, 128^3, the variable-coefficient twin at 48^3 and
96^3) and by tools/gen_golden_f5_256.py (round 5: the compiled reference run at 256^3 itself -- 14 iterations,
relres 6.3426837114e-09 as BASELINE.md section 2 recorded, now with the whole residual history and a solution sample).

Bars (north_star): equal iteration counts; |relres_gpu - relres_ref| <= 1e-10 (absolute: SURVEY section 8(d)'s parity
statement) AND <= 1e-6 * relres_ref (relative: what the device path achieves with its regrouped -- not the reference's
serial -- dot products and row sums is 1e-8 .. 2e-7; a drift beyond 1e-6 fails here long before it reaches the absolute
bar; north_star's literal "1e-10 relative" on a 6e-9 quantity would need the reference's left-to-right summation order
in every reduction of 16.8 M terms); level sizes equal; the residual history to rtol 1e-8; the
solution to 1e-9 of its maximum.  Plus the kernel A/B identities at
sizes where the full-chip kernel variants (16-bit ids, slab schedule, exception lists, 16-byte staged
streams) are the ones that run.
"""
import os

import numpy as np
import pytest

import faspsolver_amd as fa
from faspsolver_amd import _types as T

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RELRES_TOL = 1e-10   # north_star: final residual within 1e-10 of the reference's
RELRES_RTOL = 1e-6   # ... and within 1e-6 of it in relative terms (drift guard; achieved: 1e-8 .. 2e-7)
HIST_RTOL = 1e-8
X_TOL = 1e-9


def _same_history(hist, ref_hist):
    """The device history ends with two entries for the last iteration (recurrence residual, then the true residual
    b - A x the reference prints); the reference's log keeps the true one.  Entries agree to HIST_RTOL; the last one --
    a difference of O(1) vectors -- to 1e-10 of the initial residual (north_star's bar)."""
    h = np.concatenate([hist[:-2], hist[-1:]])
    return (len(h) == len(ref_hist) and np.allclose(h[:-1], ref_hist[:-1], rtol=HIST_RTOL, atol=0.0)
            and abs(h[-1] - ref_hist[-1]) <= RELRES_TOL * ref_hist[0])


def _params():
    itp = fa.param_solver_init(); itp.tol = 1e-8; itp.maxit = 500; itp.print_level = 0
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    return itp, amgp


def _levels(H):
    out = []
    for l in range(H.num_levels):
        r, c, ia, ja, v = H.matrix(l, 0)
        out.append([r, len(v)])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 128])
def test_p7_matches_reference_at_scale(gpu, n):
    z = np.load(os.path.join(G, "p7_scale.npz"))
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    assert _levels(H) == z[f"n{n}_levels"].tolist()
    st, x, hist, stats = H.solve(f, itp)
    ref_hist = z[f"n{n}_hist"]
    assert st == int(z[f"n{n}_iters"])
    assert abs(stats.relres - float(z[f"n{n}_relres"])) <= RELRES_TOL
    assert abs(stats.relres - float(z[f"n{n}_relres"])) <= RELRES_RTOL * float(z[f"n{n}_relres"])
    assert _same_history(hist, ref_hist)
    step = max(1, len(x) // 4096)
    xs = z[f"n{n}_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL * np.abs(xs).max()
    s, mx, n2 = z[f"n{n}_xsum"]
    assert abs(np.abs(x).max() - mx) <= X_TOL * mx and abs(np.sqrt((x * x).sum()) - n2) <= X_TOL * n2
    H.close()


@pytest.mark.gpu
def test_p7_256_headline_iterations_and_residual(gpu):
    """The benchmark configuration itself against the compiled reference's own run at this size (tools/gen_golden_f5_256.py):
    14 iterations, relres 6.3426837114e-09, the residual history to 1e-8, a 4096-entry sample of the solution and its norms to
    1e-9, the reference's ten level sizes -- and the plain-CSR kernels reproducing the coded ones on the same hierarchy."""
    z = np.load(os.path.join(G, "p7_scale.npz"))
    n = 256
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    assert _levels(H) == z["n256_levels"].tolist()
    H.set_rhs(f)
    st, hist, stats = H.solve_resident(itp)
    assert st == int(z["n256_iters"])
    assert abs(stats.relres - float(z["n256_relres"])) <= RELRES_TOL
    assert abs(stats.relres - float(z["n256_relres"])) <= RELRES_RTOL * float(z["n256_relres"])
    assert _same_history(hist, z["n256_hist"])
    x = H.get_solution()
    assert np.abs(x - ue).max() < 2e-5   # second-order discretisation error of the generator's exact solution
    step = max(1, len(x) // 4096)
    xs = z["n256_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL * np.abs(xs).max()
    s, mx, n2 = z["n256_xsum"]
    assert abs(np.abs(x).max() - mx) <= X_TOL * mx and abs(np.sqrt((x * x).sum()) - n2) <= X_TOL * n2
    kinds = [H.kernel_info(l, 0)[0] for l in range(H.num_levels)]
    assert kinds[0] == 6 and kinds[1] == 6   # scalar-pattern sweep on the two coded levels
    L = fa.lib()
    L.fasp_hip_tune(b"compress", 0)
    try:
        assert H.kernel_info(0, 0)[0] == 7   # 16-byte staged stream on the plain level-0 operator
        st0, hist0, stats0 = H.solve_resident(itp)
    finally:
        L.fasp_hip_tune(b"compress", 1)
    assert st0 == st and abs(stats0.relres - stats.relres) <= 1e-13
    assert np.allclose(hist0[:-1], hist[:-1], rtol=1e-9, atol=0.0) and abs(hist0[-1] - hist[-1]) <= RELRES_TOL * hist[0]
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [48, 96])
def test_variable_coefficient_matches_reference(gpu, n):
    """-div(kappa grad u), kappa of contrast 9: no row repeats, every level on the plain-CSR kernels."""
    z = np.load(os.path.join(G, "p7_scale.npz"))
    ia, ja, a, f = fa.poisson7pt_var(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    assert _levels(H) == z[f"var{n}_levels"].tolist()
    assert H.kernel_info(0, 0)[0] in (2, 7, 8, 10)   # nothing to code
    st, x, hist, stats = H.solve(f, itp)
    assert st == int(z[f"var{n}_iters"])
    assert abs(stats.relres - float(z[f"var{n}_relres"])) <= RELRES_TOL
    assert _same_history(hist, z[f"var{n}_hist"])
    step = max(1, len(x) // 4096)
    xs = z[f"var{n}_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL * np.abs(xs).max()
    H.close()


@pytest.mark.gpu
def test_coded_vs_plain_and_gen1_vs_gen2_bit_identity_128(gpu):
    """One multigrid cycle (every operator, every epilogue, no fused dots) at 128^3 through six kernel sets:
    coded / plain x the three kernel generations.  All row sums are the reference's left-to-right sums
    of the exact stored values, so the four results agree BIT FOR BIT."""
    n = 128
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    r = np.random.default_rng(5).standard_normal(len(f))
    out = {}
    try:
        for comp in (1, 0):
            for gen2 in (2, 1, 0):   # 2: + k_csr_xtile / k_csr_wstream2 on the mid levels; 1: k_csr_lstream / k_csr_rowpat4 only; 0: round-1 kernels
                L.fasp_hip_tune(b"compress", comp); L.fasp_hip_tune(b"gen2", gen2)
                out[(comp, gen2)] = H.precond(r)
            L.fasp_hip_tune(b"gen2", 2); L.fasp_hip_tune(b"xtile", 0)   # the same mid levels through k_csr_wstream2
            out[(comp, "wstream2")] = H.precond(r)
            L.fasp_hip_tune(b"xtile", 1)
    finally:
        L.fasp_hip_tune(b"compress", 1); L.fasp_hip_tune(b"gen2", 2); L.fasp_hip_tune(b"xtile", 1)
    base = out[(1, 2)]
    for k, v in out.items():
        assert np.array_equal(v, base), k
    H.close()


@pytest.mark.gpu
def test_device_row_sort_equals_host_row_sort(gpu):
    """The long-row levels keep a device copy with every row stable-sorted by column.  The sort runs on the GPU
    (k_sort_rows, bitonic over (column, position) keys); the host std::stable_sort it replaced is still there
    behind fasp_hip_tune("device_sort", 0).  Same order -> the same lane-strided sums -> identical bits.
    The ahead-of-time upload thread (setups from a million nonzeros on) is exercised by the 96^3 build."""
    n = 96
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    L = fa.lib()
    r = np.random.default_rng(6).standard_normal(len(f))
    out = []
    try:
        for dev in (1, 0):
            L.fasp_hip_tune(b"device_sort", dev)
            H = fa.AMG(ia, ja, a, amgp)
            kinds = [H.kernel_info(l, 0)[0] for l in range(H.num_levels)]
            assert 0 in kinds    # there are sub-wavefront (long-row) levels to sort
            out.append(H.precond(r))
            H.close()
    finally:
        L.fasp_hip_tune(b"device_sort", 1)
    assert np.array_equal(out[0], out[1])


@pytest.mark.gpu
@pytest.mark.parametrize("var,smoother", [(False, T.SMOOTHER_JACOBI), (True, T.SMOOTHER_JACOBI), (False, T.SMOOTHER_L1DIAG), (True, T.SMOOTHER_L1DIAG)],
                         ids=["const-jacobi", "var-jacobi", "const-l1", "var-l1"])
def test_row_window_launches_are_bit_identical(gpu, var, smoother):
    """A row-partitioned level runs every operator as three launches of the same kernel -- interior rows while the
    halo is in flight, then the two boundary windows (hierarchy.hip.h, dist_launch).  fasp_hip_tune("split_rows", k)
    issues every operator of a single-GPU hierarchy that way: each row is computed exactly as in the single launch, so
    one V-cycle agrees bit for bit (coded kernels, their exception lists, the plain stream kernels, the sub-wavefront
    kernel), and a PCG solve -- whose fused (t, p) partials are now cut differently -- agrees to rounding.
    (Round 6: whole-operator launches of the long-row levels go through the entry-parallel stream, k_csr_estream, whose row sums
    associate differently from the row kernel's that the windows keep: the bit-for-bit statement is about windows of ONE kernel, so the
    stream is switched off here; stream against row kernel, 1e-13: tests/test_gpu_estream.py and the last lines of this test.)"""
    n = 96
    if var:
        ia, ja, a, f = fa.poisson7pt_var(n)
    else:
        ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    amgp.smoother = smoother
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    r = np.random.default_rng(7).standard_normal(len(f))
    try:
        with_stream = H.precond(r)
        L.fasp_hip_tune(b"estream", 0)
        base = H.precond(r)
        assert np.abs(with_stream - base).max() <= 1e-13 * np.abs(base).max()
        st0, x0, h0, s0 = H.solve(f, itp)
        for k in (1024, 5000, 300000):
            L.fasp_hip_tune(b"split_rows", k)
            assert np.array_equal(H.precond(r), base), k
        st1, x1, h1, s1 = H.solve(f, itp)
    finally:
        L.fasp_hip_tune(b"split_rows", 0); L.fasp_hip_tune(b"estream", 1)
    assert st1 == st0 and abs(s1.relres - s0.relres) <= 1e-6 * s0.relres   # (rounding of the re-cut dot partials, carried through the iteration)
    assert np.abs(x1 - x0).max() <= 1e-10 * np.abs(x0).max()
    H.close()


@pytest.mark.gpu
def test_relative_16bit_columns_are_bit_transparent(gpu):
    """Round 5: the long-row levels (k_csr_rows) carry their column indices as 16-bit values -- absolute where the level has at most
    65536 columns (round 2), and RELATIVE to the row's smallest column where it has more but every row spans less than 65536 of them
    (levels 3 and 4 of P7(256); here level 3 of P7(176): ~83 000 rows of ~65 entries): 10 instead of 12 bytes per entry on levels that
    are nothing but the (JA, val) stream.  Same kernel, same order of the sums: one cycle and a whole solve are BIT-IDENTICAL with the
    16-bit copies switched off (fasp_hip_tune("ja16", 0): the 32-bit indices of the same device copy).
    (Round 6: whole-operator launches of these levels go through k_csr_estream, which exists for the 16-bit form only -- with
    columns relative to the CHUNK's smallest column on such levels; the statement about ONE kernel on two index widths is made with the
    stream switched off, the stream is compared with the row kernel to 1e-13.)"""
    n = 176
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    L = fa.lib()
    H = fa.AMG(ia, ja, a, amgp)
    r = np.random.default_rng(5).standard_normal(len(f))
    try:
        info = [(l, H.matrix(l, 0)[0], H.kernel_info(l, 0)) for l in range(H.num_levels - 1)]
        rel = [l for l, rows, (kind, nbytes) in info if kind == 0 and rows > 65536]
        assert rel, info   # (a sub-wavefront level with more than 65536 columns exists at this size)
        for l in rel:
            nnz = len(H.matrix(l, 0)[4])
            assert H.kernel_info(l, 0)[1] < 10.6 * nnz, (l, H.kernel_info(l, 0), nnz)   # 10 bytes per entry + row pointers + row bases: the 16-bit copy is there
        zs = H.precond(r)
        L.fasp_hip_tune(b"estream", 0)
        z1 = H.precond(r)
        assert np.abs(zs - z1).max() <= 1e-13 * np.abs(z1).max()
        st1, x1, h1, _ = H.solve(f, itp)
        L.fasp_hip_tune(b"ja16", 0)
        for l in rel:
            assert H.kernel_info(l, 0)[1] > 11.9 * len(H.matrix(l, 0)[4])
        z0 = H.precond(r)
        st0, x0, h0, _ = H.solve(f, itp)
    finally:
        L.fasp_hip_tune(b"ja16", 1); L.fasp_hip_tune(b"estream", 1)
        H.close()
    assert np.all(np.isfinite(z1)) and np.array_equal(z1, z0)
    assert st1 == st0 and np.array_equal(x1, x0) and np.array_equal(np.asarray(h1), np.asarray(h0))


@pytest.mark.gpu
def test_host_wait_reductions_are_bit_transparent(gpu):
    """Round 5: a PCG iteration waits for the host twice instead of four times -- (z, r), beta and alpha stay on the device
    (pcg_dev_beta), the true residual of the coarse safe CG's Check III is queued behind the persistent kernel (spcg_spec), (t,p) and
    (z,r) are summed from their per-block partials by the kernels that divide by them (pcg_fold), the event pair around t = A p
    brackets every fourth launch (ev_every) -- and the chain form of the sequential sweeps has its band planes touched ahead by a
    workgroup of its own (seq_chain_touch), and the parallel pass of a sweep that starts from the zero vector reads b only
    (seq_zero_skip: the products it would have subtracted are zeros).  None of it changes a value: same divisions, same summation orders, same kernels on the
    same data.  Jacobi and GS-CF solves of P7(128) with everything off against everything on: identical iterates and histories."""
    n = 128
    ia, ja, a, f, ue = fa.poisson7pt(n)
    L = fa.lib()
    keys = ((b"pcg_dev_beta", 1), (b"spcg_spec", 1), (b"pcg_fold", 1), (b"ev_every", 4), (b"seq_chain_touch", 8), (b"seq_chain_touch_t1", 1), (b"seq_zero_skip", 1))
    try:
        for jac in (True, False):
            itp, amgp = _params() if jac else _gs_params(T.SMOOTHER_GS, 1)
            H = fa.AMG(ia, ja, a, amgp)
            out = []
            for on in (0, 1, 0):
                for k, v in keys:
                    L.fasp_hip_tune(k, v if on else (1 if k == b"ev_every" else 0))
                st, x, hist, stats = H.solve(f, itp)
                out.append((st, x, np.asarray(hist), stats.relres))
            H.close()
            assert out[0][0] == out[1][0] == out[2][0] > 0
            assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][1], out[2][1]), jac
            assert np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3], jac
    finally:
        for k, v in keys:
            L.fasp_hip_tune(k, v)


def _gs_params(smoother, order, w=1.0):
    itp = fa.param_solver_init(); itp.tol = 1e-8
    amgp = fa.param_amg_init()
    amgp.smoother = smoother; amgp.smooth_order = order; amgp.relaxation = w
    return itp, amgp


@pytest.mark.gpu
@pytest.mark.parametrize("smoother,order,w", [(T.SMOOTHER_GS, 1, 1.0), (T.SMOOTHER_GS, 0, 1.0), (T.SMOOTHER_SOR, 0, 1.1)],
                         ids=["GS-CF", "GS-natural", "SOR-natural"])
def test_sequential_sweep_kernels_agree_bit_for_bit(gpu, smoother, order, w):
    """The triangular solve of a sequential sweep (seq_split.hip.h) runs as a dataflow over strips of the sweep sequence
    (k_tri_flow: values in LDS inside a strip, through W in memory between strips, a value is its own flag) or as one
    launch per dependency class (k_tri_level).  Same slots, same row arithmetic: identical bits -- with one strip per
    level, with the strips of the default size, and with strips of 16 KB (dozens of strips per level: ghosts, importer and
    exporter waves, the ticket counter all at work)."""
    n = 40
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _gs_params(smoother, order, w)
    L = fa.lib()
    r = np.random.default_rng(11).standard_normal(len(f))
    out = []
    try:
        # spine (seq_sched.h: a row's last two operands in its last lane, after the cross-lane sum) is part of a schedule's
        # arithmetic: 1 = where the schedule chooses it (the default), 2 = on every level with two lanes per row or more
        for spine in (1, 2):
            L.fasp_hip_tune(b"seq_spine", spine)
            for flow, kb in ((1, 512), (0, 512), (1, 16), (1, 1 << 20)):
                L.fasp_hip_tune(b"seq_flow", flow); L.fasp_hip_tune(b"seq_strip_kb", kb)
                H = fa.AMG(ia, ja, a, amgp)    # (the strips are cut when a level's schedule is built: one hierarchy per setting)
                out.append(H.precond(r))
                out.append(H.precond(out[-1]))
                H.close()
    finally:
        L.fasp_hip_tune(b"seq_flow", 1); L.fasp_hip_tune(b"seq_strip_kb", 0); L.fasp_hip_tune(b"seq_spine", 1)
    assert np.all(np.isfinite(out[0]))
    for base in (0, 8):   # the four settings of one spine mode agree bit for bit
        for k in range(base + 2, base + 8, 2):
            assert np.array_equal(out[base], out[k]) and np.array_equal(out[base + 1], out[k + 1])
    assert np.allclose(out[0], out[8], rtol=1e-10, atol=1e-13 * np.abs(out[0]).max())   # the two modes: the same sweep, rounded differently


@pytest.mark.gpu
def test_renumbered_hierarchy_refuses_order_dependent_parameters(gpu):
    """the brick numbering of the mid levels is decided once, at upload, from the parameters of the SETUP.  A later call that
    brings an order-dependent smoother / a recursive cycle / coarse scaling (fasp_hip_amg_solve's per-call AMG_param) would apply sweep
    schedules, C/F markers and polynomial diagonals built in natural order to brick-ordered vectors: it is refused with
    ERROR_INPUT_PAR instead of answering wrongly; the same call on a hierarchy uploaded with renumber = 0 runs and agrees with a
    hierarchy that was set up for that smoother."""
    ia, ja, a, f = fa.poisson7pt_var(40)
    L = fa.lib()
    itp, amgp = _params()
    gs = fa.param_amg_init(); gs.smoother = T.SMOOTHER_GS; gs.maxit = 30; gs.tol = 1e-8; gs.print_level = 0
    jac = fa.param_amg_init(); jac.smoother = T.SMOOTHER_JACOBI; jac.relaxation = 0.6667; jac.maxit = 60; jac.tol = 1e-8; jac.print_level = 0
    H = fa.AMG(ia, ja, a, amgp)
    try:
        assert H.num_levels >= 4
        st, x, hist, stats = H.amg_solve(f, jac)          # the setup's own kind of smoother: fine
        assert st > 0
        for bad in ("smoother", "cycle", "scaling"):
            q = fa.param_amg_init(); q.smoother = T.SMOOTHER_JACOBI; q.relaxation = 0.6667; q.maxit = 5; q.print_level = 0
            if bad == "smoother": q.smoother = T.SMOOTHER_GS
            if bad == "cycle": q.cycle_type = 3          # AMLI
            if bad == "scaling": q.coarse_scaling = 1
            st, x, hist, stats = H.amg_solve(f, q)
            assert st == T.ERROR_INPUT_PAR, (bad, st)
    finally:
        H.close()
    L.fasp_hip_tune(b"renumber", 0)
    try:
        H = fa.AMG(ia, ja, a, amgp)
        st0, x0, hist0, stats0 = H.amg_solve(f, gs)
        H.close()
    finally:
        L.fasp_hip_tune(b"renumber", 1)
    Hg = fa.AMG(ia, ja, a, gs)
    st1, x1, hist1, stats1 = Hg.amg_solve(f, gs)
    Hg.close()
    assert st0 == st1 and st0 > 0
    assert np.abs(x0 - x1).max() <= 1e-12 * np.abs(x1).max()


@pytest.mark.gpu
@pytest.mark.parametrize("variable", [False, True], ids=["P7", "variable-kappa"])
@pytest.mark.parametrize("smoother", [T.SMOOTHER_JACOBI, T.SMOOTHER_L1DIAG], ids=["jacobi", "l1"])
def test_renumbered_levels_are_bit_transparent(gpu, variable, smoother):
    """Round 5: the uncoded mid levels are renumbered in breadth-first balls of 64 rows at upload (csrc/reorder.cpp, hierarchy.hip.h:
    A_l, R_l, P_l permuted, the level's vectors live in the new order; level 0 and the coarsest level keep theirs) when the smoother
    does not depend on the order of the rows.  The stream kernels (levels of up to 48 entries per row) sum a row in its storage order:
    their results are BIT-IDENTICAL with and without it.  The sub-wavefront kernel of the long-row levels works on a device copy whose
    rows are sorted by COLUMN -- it never followed the storage order -- so there the association of a row sum follows the numbering:
    the same products, differences of an ulp.  Hence: one cycle (V and W) to 1e-13 of its largest entry, equal iteration counts, residual
    histories to 1e-9, solutions to 1e-11 -- with the default chunks and with small ones (balls across many seams); and the parity pins of
    this file (reference runs at 64^3 .. 256^3) hold with the renumbering on, which is the default."""
    n = 56
    ia, ja, a, f, ue = fa.poisson7pt(n)
    if variable:
        ia, ja, a, f = fa.poisson7pt_var(n)
    L = fa.lib()
    r = np.random.default_rng(23).standard_normal(len(f))
    out = {}
    kinds = {}
    try:
        for cyc in (1, 2):
            for ren, chunk in ((1, 262144), (0, 262144), (1, 4096), (2, 262144)):
                itp, amgp = _params()
                amgp.smoother = smoother; amgp.cycle_type = cyc
                L.fasp_hip_tune(b"renumber", ren); L.fasp_hip_tune(b"renumber_chunk", chunk)
                H = fa.AMG(ia, ja, a, amgp)
                z1 = H.precond(r)
                st, x, hist, stats = H.solve(f, itp)
                out[(cyc, ren, chunk)] = (z1, st, x, hist)
                kinds[(cyc, ren, chunk)] = [H.kernel_info(l, 0)[0] for l in range(H.num_levels)]
                H.close()
    finally:
        L.fasp_hip_tune(b"renumber", 1); L.fasp_hip_tune(b"renumber_chunk", 262144)
    for cyc in (1, 2):
        base = out[(cyc, 0, 262144)]
        assert np.all(np.isfinite(base[0])) and base[1] > 0
        for key in ((cyc, 1, 262144), (cyc, 1, 4096), (cyc, 2, 262144)):   # (2: also behind the coded level 1, through a numbering bridge)
            z1, st, x, hist = out[key]
            assert np.abs(z1 - base[0]).max() <= 1e-13 * np.abs(base[0]).max(), key
            assert st == base[1] and np.allclose(hist[:-1], base[3][:-1], rtol=1e-9, atol=0.0), key
            assert np.abs(x - base[2]).max() <= 1e-11 * np.abs(base[2]).max(), key
        assert not np.array_equal(out[(cyc, 1, 262144)][0], base[0]) or not variable   # (something really was renumbered)
    assert len(kinds[(1, 1, 262144)]) >= 5   # (levels 1 .. n-2 exist: something was there to renumber)


@pytest.mark.gpu
@pytest.mark.parametrize("smoother,order,w", [(T.SMOOTHER_GS, 1, 1.0), (T.SMOOTHER_GS, 0, 1.0), (T.SMOOTHER_SOR, 0, 1.1)],
                         ids=["GS-CF", "GS-natural", "SOR-natural"])
def test_chain_form_kernels_agree_bit_for_bit(gpu, smoother, order, w):
    """Round 5: on chain-bound sweeps the triangular solve is a blocked substitution whose dependency chain stays inside one wavefront
    (csrc/seq_chain.hip.h): k_tri_chain -- chain wave, exporter, tier-1 helpers through an LDS ring, tier-2 workgroups through memory,
    every hand-off a polled sentinel -- against k_tri_chain_ref, the same arithmetic by ONE wavefront without any polling: identical
    bits, with the tiers as the schedule cuts them and with one block of tier 1 (most entries in tier 2).  Forced onto every level
    where the form applies (seq_chain = 2); against the dataflow form (seq_chain = 0) the same sweep rounded differently."""
    n = 40
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _gs_params(smoother, order, w)
    L = fa.lib()
    r = np.random.default_rng(11).standard_normal(len(f))
    out = {}
    try:
        for chain, n1, ref in ((2, 0, 0), (2, 0, 1), (2, 1, 0), (2, 1, 1), (0, 0, 0)):
            L.fasp_hip_tune(b"seq_chain", chain); L.fasp_hip_tune(b"seq_chain_n1", n1); L.fasp_hip_tune(b"seq_chain_ref", ref)
            H = fa.AMG(ia, ja, a, amgp)    # (the form is chosen when a level's schedule is built: one hierarchy per setting)
            z1 = H.precond(r)
            out[(chain, n1, ref)] = (z1, H.precond(z1))
            H.close()
    finally:
        L.fasp_hip_tune(b"seq_chain", 1); L.fasp_hip_tune(b"seq_chain_n1", 0); L.fasp_hip_tune(b"seq_chain_ref", 0)
    assert np.all(np.isfinite(out[(2, 0, 0)][0]))
    for n1 in (0, 1):
        for k in (0, 1):
            assert np.array_equal(out[(2, n1, 0)][k], out[(2, n1, 1)][k]), (n1, k)
    big = np.abs(out[(0, 0, 0)][0]).max()
    for key in ((2, 0, 0), (2, 1, 0)):
        assert np.allclose(out[key][0], out[(0, 0, 0)][0], rtol=1e-10, atol=1e-13 * big)
    assert not np.array_equal(out[(2, 0, 0)][0], out[(0, 0, 0)][0])   # (the chain form really ran: another association of the row sums)


@pytest.mark.gpu
def test_chain_form_time_out_fails_loudly_and_falls_back(gpu):
    """Every poll of the chain form is bounded (two seconds): a launch whose tier-2 workgroups never come (fasp_hip_tune("seq_test_hang", 1):
    the hook of this test) ends, the error word -- in host-mapped memory, written by the kernels themselves -- fails THAT application of the
    preconditioner with an error code, and later sweeps take the plain one-wavefront form: the next application gives what the plain
    form gives.  Run in a process of its own (the fallback is process-wide until fasp_hip_tune("seq_flow", 1) re-enables the dataflow forms)."""
    code = r"""
import numpy as np, faspsolver_amd as fa
from faspsolver_amd import _types as T
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(40)
amgp = fa.param_amg_init()
r = np.random.default_rng(11).standard_normal(len(f))
L.fasp_hip_tune(b"seq_chain", 2); L.fasp_hip_tune(b"seq_chain_n1", 1)
L.fasp_hip_tune(b"seq_chain_ref", 1)
H = fa.AMG(ia, ja, a, amgp); zref = H.precond(r); H.close()          # what the plain form gives
L.fasp_hip_tune(b"seq_chain_ref", 0)
H = fa.AMG(ia, ja, a, amgp)
z0 = H.precond(r)
assert np.array_equal(z0, zref)
L.fasp_hip_tune(b"seq_test_hang", 1)
try:
    H.precond(r)
    print("NO-ERROR")
except RuntimeError as e:
    print("FAILED-AS-EXPECTED", e)
L.fasp_hip_tune(b"seq_test_hang", 0)
z1 = H.precond(r)                                                      # plain form from here on
print("FALLBACK-EQUAL", bool(np.array_equal(z1, zref)))
L.fasp_hip_tune(b"seq_flow", 1)                                         # re-enabled
z2 = H.precond(r)
print("REENABLED-EQUAL", bool(np.array_equal(z2, zref)))
H.close()
"""
    import subprocess, sys
    env = dict(os.environ); env["PYTHONPATH"] = os.path.dirname(G.rstrip("/")).rsplit("/tests", 1)[0] + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "FAILED-AS-EXPECTED" in p.stdout and "NO-ERROR" not in p.stdout, p.stdout
    assert "FALLBACK-EQUAL True" in p.stdout and "REENABLED-EQUAL True" in p.stdout, p.stdout
    assert "timed out" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("tag,smoother,order,w", [("gscf", T.SMOOTHER_GS, 1, 1.0), ("gsnat", T.SMOOTHER_GS, 0, 1.0), ("sor11", T.SMOOTHER_SOR, 0, 1.1)])
def test_chain_form_everywhere_matches_reference_64(gpu, tag, smoother, order, w):
    """The chain form forced onto EVERY level it applies to (up to 65 407 rows: levels 1.. of P7(64)), tier 1 cut to two blocks: the
    reference's iteration counts and residuals (tests/golden/p7_sweeps.npz) to the same bars as the default schedules."""
    z = np.load(os.path.join(G, "p7_sweeps.npz"))
    ia, ja, a, f, ue = fa.poisson7pt(64)
    itp, amgp = _gs_params(smoother, order, w)
    L = fa.lib()
    try:
        L.fasp_hip_tune(b"seq_chain", 2); L.fasp_hip_tune(b"seq_chain_n1", 2)
        H = fa.AMG(ia, ja, a, amgp)
        st, x, hist, stats = H.solve(f, itp)
        H.close()
    finally:
        L.fasp_hip_tune(b"seq_chain", 1); L.fasp_hip_tune(b"seq_chain_n1", 0)
    assert st == int(z[f"{tag}_iters"]), st
    assert abs(stats.relres - float(z[f"{tag}_relres"])) <= RELRES_TOL
    assert _same_history(hist, z[f"{tag}_hist"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,smoother,order,w", [("gsnat", T.SMOOTHER_GS, 0, 1.0), ("sor11", T.SMOOTHER_SOR, 0, 1.1), ("gscf", T.SMOOTHER_GS, 1, 1.0)])
def test_sweeps_with_many_virtual_rows_match_reference(gpu, tag, smoother, order, w):
    """Four lanes per row forced (fasp_hip_tune("seq_lanes", 4)): a work item holds 32 lower entries, so every row of the coarse
    levels of P7(64) -- 60 to 400 entries -- hands its oldest entries to up to a dozen VIRTUAL ROWS (csrc/seq_sched.h), with and
    without the spine.  Another association of every row sum, the same sweep: the reference's iteration count and residual
    (tests/golden/p7_sweeps.npz) to the same bars as the default schedule."""
    z = np.load(os.path.join(G, "p7_sweeps.npz"))
    ia, ja, a, f, ue = fa.poisson7pt(64)
    itp, amgp = _gs_params(smoother, order, w)
    L = fa.lib()
    try:
        for spine in (0, 2):
            L.fasp_hip_tune(b"seq_lanes", 4); L.fasp_hip_tune(b"seq_spine", spine)
            H = fa.AMG(ia, ja, a, amgp)
            st, x, hist, stats = H.solve(f, itp)
            H.close()
            assert st == int(z[f"{tag}_iters"]), (spine, st)
            assert abs(stats.relres - float(z[f"{tag}_relres"])) <= RELRES_TOL
            assert _same_history(hist, z[f"{tag}_hist"])
    finally:
        L.fasp_hip_tune(b"seq_lanes", 0); L.fasp_hip_tune(b"seq_spine", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("smoother,order,w", [(T.SMOOTHER_GS, 1, 1.0), (T.SMOOTHER_SOR, 0, 1.1)], ids=["GS-CF", "SOR-natural"])
def test_multicolour_sweep_mode_converges_and_is_deterministic(gpu, smoother, order, w):
    """fasp_hip_tune("gs_multicolor", 1) -- the FLAGGED NON-PARITY mode: rows are relaxed colour by colour (greedy
    colouring) instead of in index order.  A different Gauss-Seidel / SOR iteration: it must converge to the same
    solution in about as many PCG iterations as the reference's sweep (not in exactly as many), reproducibly, and
    the default mode must be untouched by having used it."""
    n = 48
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _gs_params(smoother, order, w)
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    try:
        st0, x0, h0, s0 = H.solve(f, itp)
        L.fasp_hip_tune(b"gs_multicolor", 1)
        st1, x1, h1, s1 = H.solve(f, itp)
        st2, x2, h2, s2 = H.solve(f, itp)
        L.fasp_hip_tune(b"gs_multicolor", 0)
        st3, x3, h3, s3 = H.solve(f, itp)
    finally:
        L.fasp_hip_tune(b"gs_multicolor", 0)
    assert st0 > 0 and st1 > 0 and abs(st1 - st0) <= 3
    assert s1.relres <= 1e-8
    assert np.abs(x1 - x0).max() <= 1e-6 * np.abs(x0).max()      # same solution, to the solver tolerance
    assert st2 == st1 and np.array_equal(x1, x2)                 # reproducible
    assert st3 == st0 and np.array_equal(x3, x0)                 # parity mode unchanged afterwards
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("var", [False, True])
def test_fused_zr_product_matches_separate_dot(gpu, var):
    """PCG takes (z, r) from the partial sums the last Jacobi sweep of level 0 leaves behind (x_new . b, row by row as
    the sweep writes x_new) instead of a separate pass over z and r.  Same products, another grouping of the partial
    sums: iteration counts equal, residuals equal to rounding.  fasp_hip_tune("fuse_zr", 0) is the separate pass."""
    n = 64
    if var:
        ia, ja, a, f = fa.poisson7pt_var(n)
    else:
        ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    try:
        out = []
        for fz in (1, 0):
            L.fasp_hip_tune(b"fuse_zr", fz)
            out.append(H.solve(f, itp))
    finally:
        L.fasp_hip_tune(b"fuse_zr", 1)
    (st1, x1, h1, s1), (st0, x0, h0, s0) = out
    assert st1 == st0 and len(h1) == len(h0)
    assert np.allclose(h1, h0, rtol=1e-9, atol=1e-13 * h0[0])
    assert np.abs(x1 - x0).max() <= 1e-11 * np.abs(x0).max()
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("var,cycle", [(False, 1), (True, 1), (False, 2)])
def test_fused_first_jacobi_sweep_is_bit_identical(gpu, var, cycle):
    """The first pre-smoothing sweep of a level (Jacobi from a zero guess, x = (w b) / d) is written by the kernel that
    produces b -- the restriction of the level above, on level 0 the CG update that produces r -- with the expression
    of the stand-alone kernel.  fasp_hip_tune("fuse_presmooth", 0) runs the stand-alone kernel: same bits, in a
    V-cycle apply, a W-cycle apply and a whole PCG solve."""
    n = 64
    if var:
        ia, ja, a, f = fa.poisson7pt_var(n)
    else:
        ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    amgp.cycle_type = cycle
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    r = np.random.default_rng(13).standard_normal(len(f))
    try:
        out = []
        for fp in (1, 0):
            L.fasp_hip_tune(b"fuse_presmooth", fp)
            out.append((H.precond(r), H.solve(f, itp)))
    finally:
        L.fasp_hip_tune(b"fuse_presmooth", 1)
    (z1, (st1, x1, h1, s1)), (z0, (st0, x0, h0, s0)) = out
    assert np.array_equal(z1, z0)
    assert st1 == st0 and np.array_equal(h1, h0) and np.array_equal(x1, x0)
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("var", [False, True])
def test_xtile_kernel_is_bit_identical_to_wstream2(gpu, var):
    """k_csr_xtile stages the distinct x entries of a 64-row tile in LDS and reads 16-bit column positions; it is built
    only for operators whose tiles share their columns (>= 2.5 entries per distinct column).  The coarse levels of this
    hierarchy share less, so the threshold is lowered for the test: the levels that then run it must give the bits
    k_csr_wstream2 gives (same products, same left-to-right sums)."""
    n = 64
    if var:
        ia, ja, a, f = fa.poisson7pt_var(n)
    else:
        ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    L = fa.lib()
    old = os.environ.get("FASP_HIP_XTILE_MIN_SHARE")
    os.environ["FASP_HIP_XTILE_MIN_SHARE"] = "1.0"
    try:
        H = fa.AMG(ia, ja, a, amgp)
    finally:
        if old is None:
            del os.environ["FASP_HIP_XTILE_MIN_SHARE"]
        else:
            os.environ["FASP_HIP_XTILE_MIN_SHARE"] = old
    kinds = [H.kernel_info(l, w)[0] for l in range(H.num_levels - 1) for w in (0, 1, 2)]
    assert 10 in kinds, kinds          # some operator really runs k_csr_xtile
    r = np.random.default_rng(17).standard_normal(len(f))
    try:
        z1 = H.precond(r)
        st1, x1, h1, s1 = H.solve(f, itp)
        L.fasp_hip_tune(b"xtile", 0)
        z0 = H.precond(r)
        st0, x0, h0, s0 = H.solve(f, itp)
    finally:
        L.fasp_hip_tune(b"xtile", 1)
    assert np.array_equal(z1, z0)
    assert st1 == st0 and np.array_equal(h1, h0) and np.array_equal(x1, x0)
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,fixture", [(64, "p7_sweeps.npz"), (128, "p7_sweeps_128.npz")], ids=["64", "128"])
@pytest.mark.parametrize("tag,smoother,order,w", [("gscf", T.SMOOTHER_GS, 1, 1.0), ("gsnat", T.SMOOTHER_GS, 0, 1.0), ("sor11", T.SMOOTHER_SOR, 0, 1.1)])
def test_sequential_smoothers_match_reference_64(gpu, tag, smoother, order, w, n, fixture):
    """The parity mode of the sequential smoothers at 64^3 and 128^3 against the REFERENCE's own runs
    (tests/golden/p7_sweeps*.npz, tools/gen_golden_sweeps.py): Gauss-Seidel in C/F order (the reference's defaults), in
    natural order, SOR(1.1).  The triangular solves run as a dataflow over strips of the sweep sequence (csrc/seq_split.hip.h):
    one or two strips per deep level at 64^3, hundreds per wide level (more strips than workgroups resident) at 128^3: equal
    iteration counts, residual histories to 1e-8, |relres - ref| <= 1e-10."""
    z = np.load(os.path.join(G, fixture))
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _gs_params(smoother, order, w)
    H = fa.AMG(ia, ja, a, amgp)
    st, x, hist, stats = H.solve(f, itp)
    assert st == int(z[f"{tag}_iters"])
    assert abs(stats.relres - float(z[f"{tag}_relres"])) <= RELRES_TOL
    assert _same_history(hist, z[f"{tag}_hist"])
    step = max(1, len(x) // 4096)
    xs = z[f"{tag}_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL * np.abs(xs).max()
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,smoother,order,w", [("gscf", T.SMOOTHER_GS, 1, 1.0), ("sor11", T.SMOOTHER_SOR, 0, 1.1), ("gsnat", T.SMOOTHER_GS, 0, 1.0)])
def test_sequential_smoothers_match_reference_256(gpu, tag, smoother, order, w):
    """The same at the size of the metric, P7(256): the reference's default smoother (12 iterations, relres 7.9576097848e-09),
    SOR(1.1) in natural order (10 iterations, 3.0722787533e-09) and, round 5, Gauss-Seidel in natural order (11 iterations,
    1.3410360094e-09: the t * (1 / a_ii) form of fasp_smoother_dcsr_gs) -- tests/golden/p7_sweeps_256.npz from the compiled reference
    (tools/gen_golden_sweeps_256.py).  Levels 0-4 run the dataflow form (level 0 in natural order: more strips than are resident),
    levels 5-8 the chain form (csrc/seq_chain.hip.h)."""
    z = np.load(os.path.join(G, "p7_sweeps_256.npz"))
    ia, ja, a, f, ue = fa.poisson7pt(256)
    itp, amgp = _gs_params(smoother, order, w)
    H = fa.AMG(ia, ja, a, amgp)
    H.set_rhs(f)
    st, hist, stats = H.solve_resident(itp)
    assert st == int(z[f"{tag}_iters"])
    assert abs(stats.relres - float(z[f"{tag}_relres"])) <= RELRES_TOL
    assert _same_history(hist, z[f"{tag}_hist"])
    x = H.get_solution()
    step = max(1, len(x) // 4096)
    xs = z[f"{tag}_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL * np.abs(xs).max()
    H.close()


# --- full-size pins of BASELINE.json configs 3, 4 (one GPU) and 5 ---------------------------------------------------
# tests/golden/configs_full.npz comes from the REFERENCE ITSELF (oracle/_ref/libfasp_ref.so) through
# tools/gen_golden_configs.py: iteration counts, the true relative residual ||b - A x|| / ||b|| of the reference's
# solution, checksums and a strided sample of x.  Bars: equal iteration counts, |relres_gpu - relres_ref| <= 1e-10
# (both computed the same way, on the host, from the returned x), the solution to 1e-8 of its maximum (a Krylov
# solution at rtol 1e-8 after 66 / 89 iterations: regrouped device reductions move the last digits of every
# Arnoldi coefficient; the iteration counts are the sharp test).
X_TOL_FULL = 1e-8


def _true_relres_csr(ia, ja, a, x, f):
    import scipy.sparse as sp
    A = sp.csr_matrix((a, ja, ia), shape=(len(f), len(f)))
    r = f - A @ x
    return float(np.sqrt(r @ r) / np.sqrt(f @ f))


@pytest.mark.gpu
def test_config3_full_size_matches_reference(gpu):
    """P7(128) (x) B3, UA-AMG (VMB) + block Jacobi + VGMRES(30): 66 iterations as the compiled reference."""
    import scipy.sparse as sp
    from _libs import B3, bsr_params
    z = np.load(os.path.join(G, "configs_full.npz"))
    n = int(z["c3_n"]); nb = 3
    ia, ja, a, f0, ue = fa.poisson7pt(n)
    val = (a[:, None, None] * B3[None, :, :]).reshape(-1)
    f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
    itp, amgp = bsr_params(5)
    Gh = fa.BSRAMG(ia, ja, val, nb, amgp)
    st, x, hist, stats = Gh.solve(f, itp)
    Gh.free()
    assert st == int(z["c3_iters"]) == 66
    A = sp.csr_matrix((a, ja, ia), shape=(len(ia) - 1, len(ia) - 1))
    r = f.reshape(-1, nb) - (A @ x.reshape(-1, nb)) @ B3.T
    rr = float(np.sqrt((r * r).sum()) / np.sqrt(f @ f))
    assert abs(rr - float(z["c3_relres_true"])) <= RELRES_TOL
    step = max(1, len(x) // 4096)
    xs = z["c3_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL_FULL * np.abs(xs).max()
    s, mx, n2 = z["c3_xsum"]
    assert abs(np.sqrt((x * x).sum()) - n2) <= X_TOL_FULL * n2


@pytest.mark.gpu
def test_config5_full_size_matches_reference(gpu):
    """Q1 27-point anisotropic (1, 1, 0.01) at n = 123 (49.4 M nnz), SA-AMG + W-cycle + w-Jacobi + VFGMRES(30):
    89 iterations as the compiled reference."""
    z = np.load(os.path.join(G, "configs_full.npz"))
    n = int(z["c5_n"])
    ia, ja, a, f = fa.aniso27pt(n)
    assert [len(f), len(a)] == z["c5_shape"].tolist()
    itp = fa.param_solver_init(); amgp = fa.param_amg_init()
    itp.tol = 1e-8; itp.itsolver_type = 6; itp.restart = 30
    amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE
    amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    H = fa.AMG(ia, ja, a, amgp)
    st, x, hist, stats = H.solve(f, itp)
    H.close()
    assert st == int(z["c5_iters"]) == 89
    rr = _true_relres_csr(ia, ja, a, x, f)
    assert abs(rr - float(z["c5_relres_true"])) <= RELRES_TOL
    step = max(1, len(x) // 4096)
    xs = z["c5_xsample"]
    assert np.abs(x[::step] - xs).max() <= X_TOL_FULL * np.abs(xs).max()


@pytest.mark.gpu
def test_config4_p7_512_on_one_gpu(gpu):
    """Config 4's system on ONE GPU: P7(512), 134 M DOF -- 31 iterations (the oracle's count on the same hierarchy,
    profiles/r03_check512_single_gpu.txt), the exact solution of the generator to discretisation-free 1e-5.
    FASP_TEST_512: "1" (the default) RUNS it and FAILS -- not skips -- when the host cannot hold the setup (~55 GB) or the
    run takes longer than FASP_TEST_512_BUDGET_S (600 s): config 4 must not drop out of a driver run quietly; "0" switches the
    test off on purpose (development hosts with little memory), which the skip reason then says."""
    import time
    mode = os.environ.get("FASP_TEST_512", "1")
    if mode == "0":
        pytest.skip("switched off on purpose (FASP_TEST_512=0): 2 minutes of host setup, 55 GB of host memory")
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
    except Exception:
        avail_kb = 1 << 40
    assert avail_kb >= 70 * (1 << 20), (f"P7(512) needs ~55 GB of host memory for the setup, {avail_kb >> 20} GB available: config 4 cannot be "
                                        "checked on this host (FASP_TEST_512=0 switches the test off deliberately)")
    t_start = time.perf_counter()
    n = 512
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp, amgp = _params()
    H = fa.AMG(ia, ja, a, amgp)
    H.set_rhs(f)
    st, hist, stats = H.solve_resident(itp)
    x = H.get_solution()
    H.close()
    assert st == 31
    assert stats.relres <= 1e-8 and abs(stats.relres - 6.7071735872e-09) <= 1e-10
    assert np.max(np.abs(x - ue)) <= 1e-5
    took = time.perf_counter() - t_start
    assert took <= float(os.environ.get("FASP_TEST_512_BUDGET_S", "600")), f"P7(512) took {took:.0f} s (setup + solve): too slow for the driver's GPU step"
