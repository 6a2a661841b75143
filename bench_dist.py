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
    backend = os.environ.get("BENCH_COMM", "rccl")
    ndev = L.fasp_hip_device_count()
    if ndev <= 0:
        B.log("bench_dist: no HIP device")
        sys.exit(2)
    if backend == "shm":   # validation transport: the ranks may share devices
        os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"] = "1"
    dev = local_rank % ndev if backend == "shm" else local_rank
    st = L.fasp_hip_set_device(dev)
    assert st == 0, f"set_device({dev}) -> {st}"
    if backend == "shm":
        name = f"fasp_bench_{os.environ.get('MASTER_PORT', '0')}"
        st = L.fasp_hip_comm_init_shm(rank, world, name.encode())
    else:
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            st = L.fasp_hip_comm_unique_id(idbuf)
            assert st == 0, f"ncclGetUniqueId -> {st}"
        t = torch.tensor(list(idbuf.raw), dtype=torch.uint8)
        dist.broadcast(t, 0)
        ids = bytes(t.tolist())
        st = L.fasp_hip_comm_init(rank, world, ids)
    assert st == 0, f"comm init -> {st}"

    n = args.n
    # ONE host setup per node: rank 0 generates the system, runs the AMG setup and publishes
    # the host hierarchy in a shared-memory segment; the other ranks map it and upload the rows they own.
    itp, amgp = B.workload_params()
    seg = f"fasp_hier_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
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

    for _ in range(args.warmup):
        st, hist, stats = H.solve_resident(itp)
    L.fasp_hip_device_synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    spmv_ms = []
    for _ in range(args.steps):
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
    x = H.get_solution()  # own rows filled, the rest zero
    xt = torch.from_numpy(x)
    dist.all_reduce(xt)   # disjoint row ranges: sum assembles the global solution
    err = float(np.max(np.abs(xt.numpy() - ue)))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        nloc = info0["nloc"]
        # level-0 local SpMV of rank 0: bytes its kernel has to move (stored matrix form + local x incl. ghosts + y)
        kind, matrix_bytes = H.kernel_info(0, 0)
        kernel_ms = float(np.mean(spmv_ms))
        moved = matrix_bytes + 8.0 * (nloc + info0["nghost"]) + 8.0 * nloc
        roof = B.roofline_entry(kind, moved, kernel_ms, int(stats.spmv_launches) * args.steps)
        roof["kernel"] = "rank 0, " + roof["kernel"]
        # (traffic stays null: the PMC passes are single-GPU runs of the whole operator)
        out = {
            "metric": f"AMG-PCG solve DOF/s (3D 7-pt Poisson {n}^3, classical AMG V(1,1) w-Jacobi + PCG, rtol 1e-8)",
            "value": m * args.steps / elapsed, "unit": "DOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"P7({n}): 3-D 7-point Poisson {n}^3, {m} DOF, {nnz} nnz; PCG rtol 1e-8 + "
                                   "classical RS-AMG V(1,1), Jacobi w=0.6667; one step = one full solve; "
                                   f"1-D row partition over {world} GPUs, levels >= {info0['first_replicated']} replicated",
                       "rows": m, "nnz": nnz, "levels": H.num_levels, "parallelism": f"row-partition x{world} ({backend})"},
            "iterations": int(st), "relres": stats.relres, "setup_seconds": t_setup, "host_setup_seconds": t_host,
            "max_abs_error_vs_exact": err,
            "roofline": roof,
        }
        if not args.no_cpu_baseline:
            # the same routine as the single-GPU line: the oracle on this node's host cores, on rank 0 only, on a bounded
            # sample of the same solve (rank 0 holds the global host hierarchy it published)
            try:
                allc = max(1, min(B.host_cores(), int(os.environ.get("BENCH_CPU_THREADS", "64"))))
                cb, its_cpu, rr_cpu, hist_dev = B.cpu_baseline(H, ia, ja, a, f, int(st), hist,
                                                               float(os.environ.get("BENCH_CPU_BUDGET_S", "15")), allc)
                out["cpu_baseline"] = cb
                out["parity"] = {"iters_gpu": int(st), "iters_cpu": its_cpu, "relres_gpu": stats.relres,
                                 "relres_cpu": rr_cpu, "max_rel_dev_residual_history": hist_dev}
            except Exception as e:
                B.log(f"cpu_baseline failed: {e!r}")
                out.setdefault("cpu_baseline", None)
        print(json.dumps(out), flush=True)
    H.close()
    L.fasp_hip_comm_finalize()
    dist.barrier()
    dist.destroy_process_group()
