"""Multi-GPU leg of bench.py (one process per GPU, launched by torch.distributed.run).

torch.distributed (gloo, CPU) is used for the rendezvous only: it carries the RCCL unique
id from rank 0 to the others, the barriers around the timed region and the MAX-over-ranks
of the elapsed time.  All device work -- kernels, halo exchanges, all-reduces -- is issued
by libfasp_hip.so on its own HIP stream over its own RCCL communicator.

Strong scaling: the SAME P7(n) problem is row-partitioned over the N ranks (levels below
FASP_HIP_DIST_MIN_ROWS rows are replicated).  Every rank runs the (deterministic) host
setup itself; nothing but vectors and scalars ever crosses between ranks.
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
    # keep the per-rank OpenMP host setup from oversubscribing the node
    ia, ja, a, f, ue = fa.poisson7pt(n)
    m, nnz = len(f), len(a)
    itp, amgp = B.workload_params()
    t0 = time.perf_counter()
    H = fa.AMG(ia, ja, a, amgp)
    t_setup = time.perf_counter() - t0
    H.set_rhs(f)
    info0 = H.dist_info(0)
    if rank == 0:
        B.log(f"P7({n}) on {world} ranks: setup+upload {t_setup:.2f} s, levels {H.num_levels}, "
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
        # level-0 local SpMV: algorithmic bytes of this rank's row block
        r_, c_, lia, lja, lval = H.matrix(0, 0)
        kernel_ms = float(np.mean(spmv_ms))
        lnnz = int(round(nnz * nloc / m))
        Bl = 12 * lnnz + 4 * (nloc + 1) + 8 * (nloc + info0["nghost"]) + 8 * nloc
        achieved = Bl / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "AMG-PCG solve DOF/s (3D 7-pt Poisson 256^3, classical AMG V(1,1) w-Jacobi + PCG, rtol 1e-8)",
            "value": m * args.steps / elapsed, "unit": "DOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"P7({n}): 3-D 7-point Poisson {n}^3, {m} DOF, {nnz} nnz; PCG rtol 1e-8 + "
                                   "classical RS-AMG V(1,1), Jacobi w=0.6667; one step = one full solve; "
                                   f"1-D row partition over {world} GPUs, levels >= {info0['first_replicated']} replicated",
                       "rows": m, "nnz": nnz, "levels": H.num_levels, "parallelism": f"row-partition x{world} ({backend})"},
            "iterations": int(st), "relres": stats.relres, "setup_seconds": t_setup,
            "max_abs_error_vs_exact": err,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": B.PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / B.PEAK_HBM_GBS, "traffic": None,
                         "kernel": "level-0 local t = A p on rank 0, kernel family %d "
                                   "(2 wave-stream CSR, 4 byte-dictionary coded, 5 row-pattern coded)" % H.kernel_info(0, 0)[0],
                         "bytes_per_launch": Bl, "ms_per_launch": kernel_ms},
        }
        print(json.dumps(out), flush=True)
    H.close()
    L.fasp_hip_comm_finalize()
    dist.barrier()
    dist.destroy_process_group()
