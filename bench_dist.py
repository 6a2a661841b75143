"""Multi-GPU leg of bench.py (one process per GPU, launched by torch.distributed.run).

torch.distributed (gloo, CPU) is used for the rendezvous only: it carries the RCCL unique
id from rank 0 to the others, the barriers around the timed region and the MAX-over-ranks
of the elapsed time.  All device work -- kernels, halo exchanges, all-reduces -- is issued
by libfasp_hip.so on its own HIP stream over its own RCCL communicator.

Strong scaling: the SAME P7(n) problem is row-partitioned over the N ranks (levels below
FASP_HIP_DIST_MIN_ROWS rows are replicated).  Rank 0 runs the host setup once and publishes
the hierarchy in shared memory; the other ranks map it and upload their rows.  During the
solve nothing but vector entries and scalars crosses between ranks.
"""
import ctypes as C
import datetime
import json
import os
import sys
import time

import numpy as np


def main(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    import faspsolver_amd as fa
    from faspsolver_amd import _types as T
    import bench as B

    L = fa.lib()
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
    backend = os.environ.get("BENCH_COMM", "auto")
    tok = torch.tensor([int.from_bytes(os.urandom(4), "little") if rank == 0 else 0], dtype=torch.int64)
    dist.broadcast(tok, 0)   # (names of this run only: nothing a crashed earlier run left in /dev/shm is picked up)
    run_id = f"{os.environ.get('MASTER_PORT', '0')}_{int(tok[0]):08x}"
    from faspsolver_amd import comm_probe as P
    ndev = P.visible_gpu_count()   # (asked in a child process: this one has not touched the HIP runtime when the probe children start)
    if ndev <= 0:
        B.log("bench_dist: no HIP device")
        sys.exit(2)
    if backend == "auto":
        # Which transport carries this run is decided by trying them: every rank runs a small partitioned solve over each
        # candidate in a child process (faspsolver_amd/comm_probe.py) and the ranks take the first that passed everywhere.
        pname = f"fasp_probe_{run_id}"
        if ndev < world:
            os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"] = "1"

        def all_min(ok):
            t = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t[0])
        tp = time.perf_counter()
        backend = P.choose_transport(P.candidates(ndev, world), lambda t: P.run_child(t, rank, world, local_rank % ndev, f"{pname}_{t}"),
                                     all_min, (lambda s: B.log("bench_dist: " + s)) if rank == 0 else (lambda s: None))
        if backend is None:
            B.log("bench_dist: no transport passed its probe on every rank")
            sys.exit(2)
        if rank == 0:
            B.log(f"bench_dist: transport {backend} (probes took {time.perf_counter() - tp:.1f} s)")
    if backend == "shm" or (backend == "ipc" and ndev < world):   # validation: the ranks may share devices
        os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"] = "1"
    dev = local_rank % ndev if backend == "shm" or ndev < world else local_rank
    ndev_hip = L.fasp_hip_device_count()   # (first HIP call of this process: after the probe children)
    if ndev_hip != ndev:
        B.log(f"bench_dist: rank {rank}: the counting child saw {ndev} device(s), this process sees {ndev_hip}; using this process's")
        ndev = max(ndev_hip, 1)
        dev = local_rank % ndev if backend == "shm" or ndev < world else local_rank
    st = L.fasp_hip_set_device(dev)
    assert st == 0, f"set_device({dev}) -> {st}"
    name = f"fasp_bench_{run_id}"
    if backend == "shm":
        st = L.fasp_hip_comm_init_shm(rank, world, name.encode())
    elif backend == "ipc":   # peer windows (csrc/comm_ipc.h): hipIpc-mapped device windows, one kernel per exchange, no RCCL call
        st = L.fasp_hip_comm_init_ipc(rank, world, name.encode())
    else:
        idbuf = C.create_string_buffer(128)
        st = L.fasp_hip_comm_unique_id(idbuf) if rank == 0 else 0
        ok = torch.tensor([1 if st == 0 else 0], dtype=torch.int32)
        dist.broadcast(ok, 0)
        if int(ok[0]):
            t = torch.tensor(list(idbuf.raw), dtype=torch.uint8)
            dist.broadcast(t, 0)
            st = L.fasp_hip_comm_init(rank, world, bytes(t.tolist()))
        else:
            st = -1
        # every rank must have its communicator, or none uses RCCL: fall back to the peer-window transport together
        allok = torch.tensor([1 if st == 0 else 0], dtype=torch.int32)
        dist.all_reduce(allok, op=dist.ReduceOp.MIN)
        if not int(allok[0]):
            if st == 0:
                L.fasp_hip_comm_finalize()
            if rank == 0:
                B.log("bench_dist: RCCL communicator could not be created on every rank; falling back to the peer-window transport (BENCH_COMM=ipc)")
            backend = "ipc"
            st = L.fasp_hip_comm_init_ipc(rank, world, name.encode())
    assert st == 0, f"comm init -> {st}"

    def measure(n, steps, warmup, with_cpu):
        """One size: host setup on rank 0, partition + upload, warm-up and timed solves, the diagnostic solve; rank 0 returns the line's
        dictionary, the other ranks None.  Every rank takes the same path (collectives inside)."""
        out = None
        # ONE host setup per node: rank 0 generates the system, runs the AMG setup and publishes
        # the host hierarchy in a shared-memory segment; the other ranks map it and upload the rows they own.
        itp, amgp = B.workload_params()
        seg = f"fasp_hier_{run_id}_{n}_{os.getuid()}"
        t0 = time.perf_counter()
        if rank == 0:
            ia, ja, a, f, ue = fa.poisson7pt(n)
            m, nnz = len(f), len(a)
            H = fa.AMG(ia, ja, a, amgp, host_only=True)
            H.publish(seg)
            meta = torch.tensor([m, nnz], dtype=torch.int64)
        else:
            meta = torch.zeros(2, dtype=torch.int64)
        dist.broadcast(meta, 0)
        m, nnz = int(meta[0]), int(meta[1])
        if rank != 0:
            H = fa.AMG.attach(seg)
            f = np.empty(m); ue = np.empty(m)
        ft = torch.from_numpy(f); ut = torch.from_numpy(ue)
        dist.broadcast(ft, 0); dist.broadcast(ut, 0)
        t_host = time.perf_counter() - t0
        H.upload()              # partition + upload of this rank's rows (no collective inside)
        dist.barrier()
        if rank == 0:
            fa.AMG.unpublish(seg)
        t_setup = time.perf_counter() - t0
        H.set_rhs(f)
        info0 = H.dist_info(0)
        if rank == 0:
            B.log(f"P7({n}) on {world} ranks: one host setup {t_host:.2f} s, + partition/upload = {t_setup:.2f} s, levels {H.num_levels}, "
                  f"first replicated level {info0['first_replicated']}, rank-0 rows {info0['nloc']} (+{info0['nghost']} ghosts)")

        cst = (C.c_double * 8)()
        for _ in range(warmup):
            st, hist, stats = H.solve_resident(itp)
        L.fasp_hip_device_synchronize()
        L.fasp_hip_comm_stats(cst, 1)   # counters from here on: the timed solves
        dist.barrier()
        t0 = time.perf_counter()
        spmv_ms = []
        for _ in range(steps):
            st, hist, stats = H.solve_resident(itp)
            spmv_ms.append(stats.spmv_ms)
        L.fasp_hip_device_synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        # consistency across ranks: iteration count and residual are replicated scalars
        chk = torch.tensor([float(st), stats.relres], dtype=torch.float64)
        lo = chk.clone(); hi = chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks disagree on iteration count / residual"
        L.fasp_hip_comm_stats(cst, 1)
        per_solve = [v / steps for v in cst]
        # ONE more solve in the communicator's diagnostic mode: the stream is drained around every exchange / all-reduce /
        # all-gather, so each call's time is its own and what is left of the solve is this rank's kernels.  A breakdown of a
        # serialised solve (no overlap of the halo with the interior rows), never part of `value`.
        L.fasp_hip_comm_timing(1)
        dist.barrier()
        td = time.perf_counter()
        st_d, hist_d, stats_d = H.solve_resident(itp)
        L.fasp_hip_device_synchronize()
        t_diag = time.perf_counter() - td
        L.fasp_hip_comm_timing(0)
        L.fasp_hip_comm_stats(cst, 1)
        diag = torch.tensor([t_diag, cst[5], cst[6], cst[7], stats_d.solve_seconds], dtype=torch.float64)
        dmax = diag.clone(); dmin = diag.clone()
        dist.all_reduce(dmax, op=dist.ReduceOp.MAX); dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
        levels_info = [H.dist_info(l) for l in range(H.num_levels)]
        # where every rank sits: PCI bus ids, gathered over gloo (an 8-GPU run is only a measurement when they are all different)
        idbuf2 = C.create_string_buffer(64)
        L.fasp_hip_device_identity(idbuf2, 64)
        ids = [None] * world
        dist.all_gather_object(ids, idbuf2.value.decode())
        x = H.get_solution()  # own rows filled, the rest zero
        xt = torch.from_numpy(x)
        dist.all_reduce(xt)   # disjoint row ranges: sum assembles the global solution
        err = float(np.max(np.abs(xt.numpy() - ue)))

        if rank == 0:
            ms_per_step = 1e3 * elapsed / steps
            nloc = info0["nloc"]
            # level-0 local SpMV of rank 0: bytes its kernel has to move (stored matrix form + local x incl. ghosts + y)
            kind, matrix_bytes = H.kernel_info(0, 0)
            kernel_ms = float(np.mean(spmv_ms))
            moved = matrix_bytes + 8.0 * (nloc + info0["nghost"]) + 8.0 * nloc
            roof = B.roofline_entry(kind, moved, kernel_ms, int(stats.spmv_launches) * steps)
            roof["kernel"] = "rank 0, " + roof["kernel"]
            # (traffic stays null: the PMC passes are single-GPU runs of the whole operator)
            out = {
                "metric": f"AMG-PCG solve DOF/s (3D 7-pt Poisson {n}^3, classical AMG V(1,1) w-Jacobi + PCG, rtol 1e-8)",
                "value": m * steps / elapsed, "unit": "DOF/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"P7({n}): 3-D 7-point Poisson {n}^3, {m} DOF, {nnz} nnz; PCG rtol 1e-8 + "
                                       "classical RS-AMG V(1,1), Jacobi w=0.6667; one step = one full solve; "
                                       f"1-D row partition over {world} GPUs, levels >= {info0['first_replicated']} replicated",
                           "rows": m, "nnz": nnz, "levels": H.num_levels, "parallelism": f"row-partition x{world} ({backend})"},
                "iterations": int(st), "relres": stats.relres, "setup_seconds": t_setup, "host_setup_seconds": t_host,
                "max_abs_error_vs_exact": err,
                "roofline": roof,
                # what the N-GPU number is made of (per solve, rank 0's counts; times from ONE diagnostic solve, max / min over ranks)
                "comm": {"transport": backend, "devices": ids, "ranks_on_distinct_devices": bool(len(set(ids)) == world),
                         "rccl_ranks": world if backend == "rccl" else 0,
                         "sync_points_per_iteration": (per_solve[0] + per_solve[1] + per_solve[2]) / max(1, int(st)),
                         "per_solve": {"halo_exchanges": per_solve[0], "allreduces": per_solve[1], "allgathers": per_solve[2],
                                       "halo_MB_sent_rank0": per_solve[3] * 8e-6, "allgather_MB_rank0": per_solve[4] * 8e-6,
                                       "allreduces_per_iteration": per_solve[1] / max(1, int(st))},
                         "diagnostic_solve_ms": {"note": "one solve with the stream drained around every communicator call (serialised: no halo / interior overlap)",
                                                 "total_max": float(dmax[4]) * 1e3, "halo_exchange_max": float(dmax[1]) * 1e3, "halo_exchange_min": float(dmin[1]) * 1e3,
                                                 "allreduce_max": float(dmax[2]) * 1e3, "allreduce_min": float(dmin[2]) * 1e3,
                                                 "allgather_max": float(dmax[3]) * 1e3, "allgather_min": float(dmin[3]) * 1e3,
                                                 "kernels_and_host_rank_max": float(dmax[4] - dmin[1] - dmin[2] - dmin[3]) * 1e3},
                         "coarse_cg_iterations": int(stats.coarse_iters), "vcycles": int(stats.vcycles),
                         "levels": [{"level": l, "distributed": int(not i["replicated"]), "rows": i["nglobal"], "rank0_rows": i["nloc"],
                                     "rank0_ghosts": i["nghost"], "rank0_sends": i["nsend"]} for l, i in enumerate(levels_info)]},
            }
            # the reference's own figures for this solve where they are pinned (tests/golden/configs_full.npz): what a first run on N GPUs is read against
            pins = {256: (14, 6.3426837114e-09), 512: (31, 6.70717e-09)}
            if n in pins:
                out["parity_reference"] = {"iters_gpu": int(st), "iters_reference": pins[n][0], "relres_gpu": stats.relres, "relres_reference": pins[n][1],
                                           "ok": bool(int(st) == pins[n][0] and abs(stats.relres - pins[n][1]) <= 1e-10)}
            if with_cpu:
                # the same routine as the single-GPU line: the oracle on this node's host cores, on rank 0 only, on a bounded
                # sample of the same solve (rank 0 holds the global host hierarchy it published)
                try:
                    cb, its_cpu, rr_cpu, hist_dev = B.cpu_baseline(H, ia, ja, a, f, int(st), hist,
                                                                   float(os.environ.get("BENCH_CPU_BUDGET_S", "15")), B.baseline_candidates())
                    out["cpu_baseline"] = cb
                    out["parity"] = {"iters_gpu": int(st), "iters_cpu": its_cpu, "relres_gpu": stats.relres,
                                     "relres_cpu": rr_cpu, "max_rel_dev_residual_history": hist_dev}
                except Exception as e:
                    B.log(f"cpu_baseline failed: {e!r}")
                    out.setdefault("cpu_baseline", None)
        H.close()
        return out

    out = measure(args.n, args.steps, args.warmup, not args.no_cpu_baseline)
    # north_star quotes its scaling target (>= 3.5 x at 8 GPUs) at 512^3; the metric's own size, 256^3, is latency- and Amdahl-bound on
    # eight GPUs.  An 8-GPU run of the default size therefore carries the 512^3 solve too, INSIDE the one JSON
    # line (key "at_512"; BENCH_SCALE_512=0 skips it, =1 forces it for any N).  Decided from the environment alone: every rank agrees.
    want512 = os.environ.get("BENCH_SCALE_512", "1" if (world == 8 and args.n == 256) else "0") == "1" and args.n != 512
    if want512:
        o2 = measure(512, max(1, min(args.steps, 3)), 1, False)
        if rank == 0 and o2 is not None:
            out["at_512"] = {k: o2[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "iterations", "relres", "setup_seconds",
                                                "host_setup_seconds", "parity_reference", "roofline") if k in o2}
            out["at_512"]["note"] = ("north_star's scaling size; one MI355X solves it in 0.74 s (31 iterations, profiles/r05_check512_*: a separate "
                                     "single-GPU run, not part of this one)")
            out["at_512"]["comm"] = {k: o2["comm"][k] for k in ("transport", "per_solve", "sync_points_per_iteration", "diagnostic_solve_ms")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    L.fasp_hip_comm_finalize()
    dist.barrier()
    dist.destroy_process_group()
