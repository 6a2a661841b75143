"""The distributed solver end to end on the GPU box.  The box has ONE GPU, so the ranks
share it and talk through the host-staged shared-memory transport (comm.cpp, SHM backend);
partition, halo plans, replicated levels and the replicated host control flow are exactly
the code the RCCL transport drives.  Checked against the oracle: equal iteration count,
residual history to 1e-8, every rank's rows of x."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, name, n, min_rows, cycle, q, shared=False, transport="shm", idfile=None, amg_type=1, smoother=1, itsolver=1, tune="", smooth_order=1, relaxation=0.6667):
    try:
        os.environ["FASP_HIP_DIST_MIN_ROWS"] = str(min_rows)
        os.environ.setdefault("FASP_HIP_SHM_TIMEOUT_S", "60")   # (a rank that fails leaves its peers at a barrier: bound the wait)
        if tune:
            os.environ["FASP_HIP_TUNE"] = tune
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import faspsolver_amd as fa
        from faspsolver_amd import _types as T
        L = fa.lib()
        import time
        if transport == "rccl":   # the production transport: one device per rank, unique id through a file
            import ctypes as C
            assert L.fasp_hip_set_device(rank) == 0
            if rank == 0:
                buf = C.create_string_buffer(128)
                assert L.fasp_hip_comm_unique_id(buf) == 0
                with open(idfile + ".tmp", "wb") as fh:
                    fh.write(buf.raw)
                os.replace(idfile + ".tmp", idfile)
            t0 = time.time()
            while not os.path.exists(idfile):
                assert time.time() - t0 < 120, "no unique id from rank 0"
                time.sleep(0.05)
            ids = open(idfile, "rb").read()
            assert L.fasp_hip_comm_init(rank, world, ids) == 0
        elif transport == "ipc":   # peer windows between processes that share the box's one GPU
            assert L.fasp_hip_set_device(0) == 0
            assert L.fasp_hip_comm_init_ipc(rank, world, name.encode()) == 0
        else:
            assert L.fasp_hip_set_device(0) == 0
            assert L.fasp_hip_comm_init_shm(rank, world, name.encode()) == 0
        ia, ja, a, f, ue = fa.poisson7pt(n)
        itp = fa.param_solver_init(); itp.tol = 1e-8; itp.itsolver_type = itsolver
        amgp = fa.param_amg_init(); amgp.smoother = smoother; amgp.relaxation = relaxation; amgp.smooth_order = smooth_order
        amgp.cycle_type = cycle; amgp.AMG_type = amg_type
        if shared:   # one host setup: rank 0 publishes, the others attach
            seg = name + "_hier"
            flag = "/dev/shm/" + seg + ".ready"
            if rank == 0:
                H = fa.AMG(ia, ja, a, amgp, host_only=True)
                H.publish(seg)
                open(flag, "w").close()
            else:
                t0 = time.time()
                while not os.path.exists(flag):
                    assert time.time() - t0 < 120, "rank 0 never published"
                    time.sleep(0.02)
                H = fa.AMG.attach(seg)
            H.upload()
        else:
            H = fa.AMG(ia, ja, a, amgp)
        info = H.dist_info(0)
        st, x, hist, stats = H.solve(f, itp)
        H.close()
        L.fasp_hip_comm_finalize()
        if shared and rank == 0:
            fa.AMG.unpublish(name + "_hier")
            try:
                os.remove("/dev/shm/" + name + "_hier.ready")
            except OSError:
                pass
        q.put((rank, "ok", st, hist, x[info["row0"]:info["row0"] + info["nloc"]], info))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None, None))


def _run_ranks(world, n, min_rows, cycle, shared=False, transport="shm", amg_type=1, smoother=1, itsolver=1, tune="", smooth_order=1, relaxation=0.6667):
    import multiprocessing as mp
    import tempfile
    from _libs import T, default_params, orc_solve, poisson7pt
    ia, ja, a, f, ue = poisson7pt(n)
    itp, amgp = default_params()
    itp.tol = 1e-8; itp.itsolver_type = itsolver; amgp.smoother = smoother; amgp.relaxation = relaxation; amgp.smooth_order = smooth_order; amgp.cycle_type = cycle; amgp.AMG_type = amg_type
    s_ref, x_ref, h_ref, rr = orc_solve(ia, ja, a, f, itp, amgp)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"fasp_test_{os.getpid()}_{world}_{n}_{int(shared)}"
    idfile = os.path.join(tempfile.gettempdir(), name + ".ncclid")
    if os.path.exists(idfile):
        os.remove(idfile)
    procs = [ctx.Process(target=_worker, args=(r, world, name, n, min_rows, cycle, q, shared, transport, idfile, amg_type, smoother, itsolver, tune, smooth_order, relaxation)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
    finally:   # a failed rank must not leave its peers (or this test) waiting
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.terminate()
        if os.path.exists(idfile):
            os.remove(idfile)
    for r in res:
        assert r[1] == "ok", r[2]
    for rank, _, st, hist, xloc, info in res:
        assert info["replicated"] == 0 and info["first_replicated"] >= 1   # level 0 really is partitioned
        assert st == s_ref
        if len(h_ref):   # (the oracle records the residual history of CG only)
            assert len(hist) == len(h_ref)
            assert np.allclose(hist, h_ref, rtol=1e-8, atol=1e-12 * h_ref[0])
        ref_loc = x_ref[info["row0"]:info["row0"] + info["nloc"]]
        assert np.max(np.abs(xloc - ref_loc)) <= 1e-8 * np.max(np.abs(x_ref))
    # every rank saw bit-identical scalars
    for r in res[1:]:
        assert np.array_equal(r[3], res[0][3])


@pytest.mark.parametrize("world,n,min_rows,cycle,shared", [(2, 24, 500, 1, False), (3, 24, 2000, 1, False), (2, 20, 300, 2, False),
                                                           (4, 32, 3000, 1, False), (3, 24, 2000, 1, True), (2, 48, 3000, 1, True)])
def test_distributed_solve_matches_oracle(gpu, world, n, min_rows, cycle, shared):
    _run_ranks(world, n, min_rows, cycle, shared)


@pytest.mark.parametrize("world,n,min_rows,amg_type", [(2, 24, 300, 2), (3, 24, 300, 3)])
def test_distributed_aggregation_hierarchies_match_oracle(gpu, world, n, min_rows, amg_type):
    """SA (2) and UA (3) hierarchies carry no C/F marker: every level is cut into equal row blocks (dist_plan.cpp)."""
    _run_ranks(world, n, min_rows, 1, amg_type=amg_type)


def test_distributed_l1_smoother_matches_oracle(gpu):
    """The L1-diagonal smoother on row-partitioned levels (its sweep, too, runs interior rows beside the halo exchange)."""
    from _libs import T
    _run_ranks(2, 32, 500, 1, smoother=T.SMOOTHER_L1DIAG)


@pytest.mark.parametrize("itsolver", [2, 4, 5, 6], ids=["BiCGstab", "GMRES", "VGMRES", "VFGMRES"])
def test_distributed_other_krylov_methods_match_oracle(gpu, itsolver):
    """The other Krylov drivers on a row-partitioned level 0: they call halo(v) and then the operator, and the operator
    bundle turns that pair into one overlapped application (cycles.hip.h, csr_ops)."""
    _run_ranks(2, 24, 500, 1, itsolver=itsolver)


@pytest.mark.parametrize("world,n,min_rows,cycle,shared,kw", [
    (2, 24, 500, 1, False, {}), (3, 24, 2000, 1, True, {}), (4, 32, 3000, 2, False, {}), (2, 48, 3000, 1, True, {}),
    (2, 24, 300, 1, False, {"amg_type": 2}), (2, 24, 500, 1, False, {"itsolver": 6}),
    (3, 40, 40000, 1, False, {"tune": "coarse_mode=1,coarse_split_min=1024"})],
    ids=["2 ranks V", "3 ranks one setup", "4 ranks W", "2 ranks 48^3", "SA hierarchy", "VFGMRES", "split-work replicated levels"])
def test_peer_window_transport_matches_oracle(gpu, world, n, min_rows, cycle, shared, kw):
    """The peer-window transport (comm_ipc.h: hipIpc-mapped uncached windows, one kernel per halo exchange that stores into the
    neighbours' mailboxes, all-reduce by the same means, summed in rank order) between processes that share the box's one GPU:
    the same partition / halo / replicated-level code as over RCCL, against the oracle -- iteration counts, residual histories,
    every rank's rows of the solution, bit-identical scalars on all ranks."""
    _run_ranks(world, n, min_rows, cycle, shared, transport="ipc", **kw)


@pytest.mark.parametrize("world,n,min_rows,smoother,order,cycle", [
    (2, 24, 500, "GS", 1, 1), (3, 24, 800, "GS", 0, 1), (2, 20, 300, "SOR", 0, 2), (4, 32, 3000, "SGS", 0, 1), (2, 24, 500, "GSF", 1, 1)],
    ids=["GS C/F order (the reference's default), 2 ranks", "GS natural order, 3 ranks", "SOR, W-cycle", "SGS, 4 ranks", "F-point GS"])
def test_sequential_smoothers_on_partitioned_levels_match_oracle(gpu, world, n, min_rows, smoother, order, cycle):
    """fasp_hip_tune("seq_partition", 1): Gauss-Seidel / SOR sweeps on ROW-PARTITIONED levels -- the ranks take turns in sweep
    order, a halo exchange in front of every turn (smoothers.hip.h, seq_sweep): the reference's sequential iteration with the
    level's matrix and vectors distributed.  Against the oracle's single-process solve: iteration count, residual history,
    every rank's rows of the solution; level 0 really is partitioned."""
    from _libs import T
    sm = {"GS": T.SMOOTHER_GS, "SOR": T.SMOOTHER_SOR, "SGS": T.SMOOTHER_SGS, "GSF": T.SMOOTHER_GSF}[smoother]
    _run_ranks(world, n, min_rows, cycle, smoother=sm, tune="seq_partition=1", smooth_order=order, relaxation=1.1 if smoother == "SOR" else 0.6667)


def test_peer_windows_with_small_mailboxes_and_turn_taking_sweeps(gpu, monkeypatch):
    """Two more corners of the peer-window transport: mailboxes of 1 024 doubles (FASP_HIP_IPC_CAP) -- the level-0 halo of P7(32)
    fills one exactly, the all-gather at the first whole level goes through in pieces -- and the sequential smoothers on
    partitioned levels (ranks sweeping by turns) over it."""
    from _libs import T
    monkeypatch.setenv("FASP_HIP_IPC_CAP", "1024")
    _run_ranks(2, 32, 3000, 1, transport="ipc")
    _run_ranks(2, 40, 3000, 1, transport="ipc")   # (round 5: the level-0 halo of P7(40) is 1 600 doubles -- longer than a mailbox: sent in two pieces)
    monkeypatch.delenv("FASP_HIP_IPC_CAP")
    _run_ranks(3, 24, 800, 1, transport="ipc", smoother=T.SMOOTHER_GS, tune="seq_partition=1", smooth_order=1)


def test_rccl_transport_with_all_visible_gpus(gpu):
    """The production transport with real peers: world = number of visible GPUs (skipped on a one-GPU box, where the
    shared-memory transport above drives the same partition / halo / replicated-level code).  Halo ncclSend / ncclRecv,
    the all-gather at the replicated boundary and the scalar all-reduces against the oracle."""
    import faspsolver_amd as fa
    ndev = fa.lib().fasp_hip_device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: RCCL with peers needs at least two")
    _run_ranks(min(ndev, 8), 40, 4000, 1, shared=True, transport="rccl")


@pytest.mark.gpu
def test_rccl_entry_points_single_rank():
    """Every RCCL call of the production transport, on a one-rank communicator (one GPU here)."""
    import faspsolver_amd as fa
    assert fa.lib().fasp_hip_comm_selftest() == 0


@pytest.mark.parametrize("world,n,min_rows,cycle", [(2, 32, 20000, 1), (3, 40, 40000, 1), (4, 32, 20000, 2)])
def test_replicated_levels_with_split_work_match_oracle(world, n, min_rows, cycle):
    """fasp_hip_tune("coarse_mode", 1): the replicated levels keep their vectors on every rank but every rank applies an
    operator to its share of the rows only, an all-gather completes the result (hierarchy.hip.h, rep_launch).  Same
    iteration as the oracle's: iteration count, residual history, every rank's rows of the solution."""
    _run_ranks(world, n, min_rows, cycle, tune="coarse_mode=1,coarse_split_min=1024")


# --- block (BSR) path, config 3: block rows partitioned over the ranks (dist_plan.cpp, build_dist_plan_bsr) ---------------
def _bsr_worker(rank, world, name, n, min_rows, solver, cycle, q):
    try:
        os.environ["FASP_HIP_DIST_MIN_ROWS"] = str(min_rows)
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import faspsolver_amd as fa
        import _libs
        L = fa.lib()
        assert L.fasp_hip_set_device(0) == 0
        assert L.fasp_hip_comm_init_shm(rank, world, name.encode()) == 0
        ia, ja, val, nb = _libs.poisson7pt_bsr(n)
        f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
        itp, amgp = _libs.bsr_params(solver, cycle)
        G = fa.BSRAMG(ia, ja, val, nb, amgp)
        info = G.dist_info()
        st, x, hist, stats = G.solve(f, itp)
        G.free()
        L.fasp_hip_comm_finalize()
        lo, hi = info["row0"] * nb, (info["row0"] + info["nloc"]) * nb
        q.put((rank, "ok", st, stats.relres, x[lo:hi], info))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None, None))


@pytest.mark.parametrize("world,n,min_rows,solver,cycle", [(2, 16, 300, 5, 1), (3, 20, 500, 5, 1), (2, 16, 300, 1, 1), (4, 24, 800, 6, 2)])
def test_block_rows_partitioned_match_oracle(world, n, min_rows, solver, cycle):
    """P7(n) (x) B3, UA-AMG (VMB) + block Jacobi, Krylov methods of config 3 (VGMRES, CG, VFGMRES + W-cycle) on 2-4 ranks
    over the shared-memory transport: iteration count, final residual and every rank's rows of the solution against the
    oracle's single-process solve; level 0 really is partitioned."""
    import multiprocessing as mp
    from _libs import bsr_params, orc_bsr_solve, poisson7pt_bsr
    ia, ja, val, nb = poisson7pt_bsr(n)
    f = np.random.default_rng(1).standard_normal((len(ia) - 1) * nb)
    i1, a1 = bsr_params(solver, cycle)
    s_ref, x_ref, nl, rr_ref = orc_bsr_solve(ia, ja, val, nb, f, i1, a1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"fasp_bsr_{os.getpid()}_{world}_{n}_{solver}"
    procs = [ctx.Process(target=_bsr_worker, args=(r, world, name, n, min_rows, solver, cycle, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.terminate()
    for r in res:
        assert r[1] == "ok", r[2]
    for rank, _, st, relres, xloc, info in res:
        assert info["replicated"] == 0 and info["first_replicated"] >= 1
        assert st == s_ref
        assert abs(relres - rr_ref) <= 1e-10
        lo, hi = info["row0"] * nb, (info["row0"] + info["nloc"]) * nb
        assert np.max(np.abs(xloc - x_ref[lo:hi])) <= 1e-8 * np.max(np.abs(x_ref))


# --- the partitioned path at the size of the metric (round 4) ------------------------------------------
def _p7_256_worker(rank, world, name, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import time
        import faspsolver_amd as fa
        from faspsolver_amd import _types as T
        L = fa.lib()
        assert L.fasp_hip_set_device(0) == 0
        assert L.fasp_hip_comm_init_shm(rank, world, name.encode()) == 0
        n = 256
        itp = fa.param_solver_init(); itp.tol = 1e-8; itp.maxit = 500; itp.print_level = 0
        amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
        seg = name + "_hier"
        flag = "/dev/shm/" + seg + ".ready"
        ia, ja, a, f, ue = fa.poisson7pt(n)
        if rank == 0:   # ONE host setup: rank 0 publishes the hierarchy, the others map it (bench_dist.py's sequence)
            H = fa.AMG(ia, ja, a, amgp, host_only=True)
            H.publish(seg)
            open(flag, "w").close()
        else:
            del ia, ja, a
            t0 = time.time()
            while not os.path.exists(flag):
                assert time.time() - t0 < 600, "rank 0 never published"
                time.sleep(0.05)
            H = fa.AMG.attach(seg)
        H.upload()
        H.set_rhs(f)
        infos = [H.dist_info(l) for l in range(H.num_levels)]
        wins = [H.dist_list(l, 5).tolist() for l in range(H.num_levels)]
        kinds = [H.kernel_info(l, 0)[0] for l in range(2)]
        st, hist, stats = H.solve_resident(itp)
        x = H.get_solution()
        i0 = infos[0]
        lo, hi = i0["row0"], i0["row0"] + i0["nloc"]
        err = float(np.max(np.abs(x[lo:hi] - ue[lo:hi])))
        step = max(1, len(x) // 4096)
        idx = np.arange(0, len(x), step)
        own = (idx >= lo) & (idx < hi)
        H.close()
        L.fasp_hip_comm_finalize()
        if rank == 0:
            fa.AMG.unpublish(seg)
            try:
                os.remove(flag)
            except OSError:
                pass
        q.put((rank, "ok", st, stats.relres, np.asarray(hist), infos, wins, kinds, err, (own, x[idx[own]])))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))


@pytest.mark.skipif(os.environ.get("FASP_TEST_DIST256") == "0", reason="switched off (FASP_TEST_DIST256=0)")
def test_partitioned_p7_256_two_ranks_matches_reference(gpu):
    """BASELINE.json's metric workload through the ROW-PARTITIONED path (config 4's code at config 2's size): P7(256) cut
    into two z-slabs, two processes sharing the box's one GPU over the shared-memory transport, one published host
    setup.  Bars: the reference's 14 iterations, |relres - 6.3426837114e-09| <= 1e-10 (tests/golden/p7_scale.npz, from
    BASELINE.md section 2), residual history to 1e-8, levels 0-3 distributed, interior row windows in use on them (halo
    beside the interior rows), the row-pattern-coded level-0/1 kernels in their row-window form, every rank's rows of the solution
    against the generator's exact solution and the reference's sample."""
    import multiprocessing as mp
    z = np.load(os.path.join(ROOT, "tests", "golden", "p7_scale.npz"))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"fasp_p7_256_{os.getpid()}"
    procs = [ctx.Process(target=_p7_256_worker, args=(r, world, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=900) for _ in procs]
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.terminate()
    for r in res:
        assert r[1] == "ok", r[2]
    ref_hist = z["n256_hist"] if "n256_hist" in z.files else None
    for rank, _, st, relres, hist, infos, wins, kinds, err, (own, xs) in res:
        assert st == int(z["n256_iters"]) == 14
        assert abs(relres - float(z["n256_relres"])) <= 1e-10
        assert infos[0]["nranks"] == 2 and infos[0]["first_replicated"] >= 4      # levels 0-3 are row-partitioned
        for l in range(4):
            assert infos[l]["replicated"] == 0 and 0 < infos[l]["nloc"] < infos[l]["nglobal"] and infos[l]["nghost"] > 0
            assert wins[l][1] > wins[l][0] >= 0 and wins[l][1] - wins[l][0] >= infos[l]["nloc"] // 2   # interior window
        assert infos[0]["nloc"] == 256 ** 3 // 2 and infos[0]["nghost"] == 256 ** 2
        assert all(k == 6 for k in kinds), kinds   # a rank's rows of levels 0-1 are coded like the square operator (column offsets relative to the row; ghost columns behind the own ones): the scalar-pattern pair sweep, as on one GPU
        assert err < 2e-5                 # discretisation error of the generator's exact solution
        if "n256_xsample" in z.files:
            ref = z["n256_xsample"][own]
            assert np.abs(xs - ref).max() <= 1e-9 * np.abs(z["n256_xsample"]).max()
        if ref_hist is not None:
            h = np.concatenate([hist[:-2], hist[-1:]])
            assert len(h) == len(ref_hist) and np.allclose(h[:-1], ref_hist[:-1], rtol=1e-8, atol=0.0)
    assert np.array_equal(res[0][4], res[1][4])   # bit-identical replicated scalars on both ranks
