"""Transport selection of the multi-GPU bench (faspsolver_amd/comm_probe.py): the choice logic on the CPU, the probe
itself -- one child process per rank, a small partitioned solve against the same rank's unpartitioned one -- on the GPU
box (ranks sharing its one GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faspsolver_amd import comm_probe as P  # noqa: E402


def test_candidates_need_a_device_per_rank_for_rccl():
    assert P.candidates(8, 8) == ["ipc", "rccl", "shm"]
    assert P.candidates(8, 2) == ["ipc", "rccl", "shm"]
    assert P.candidates(1, 2) == ["ipc", "shm"]


@pytest.mark.parametrize("codes,expect,asked", [
    ({"ipc": [0, 0], "rccl": [0, 0], "shm": [0, 0]}, "ipc", ["ipc"]),
    ({"ipc": [0, 1], "rccl": [0, 0], "shm": [0, 0]}, "rccl", ["ipc", "rccl"]),          # one rank failing is enough
    ({"ipc": [124, 124], "rccl": [2, 0], "shm": [0, 0]}, "shm", ["ipc", "rccl", "shm"]),
    ({"ipc": [3, 3], "rccl": [1, 1], "shm": [2, 2]}, None, ["ipc", "rccl", "shm"]),
])
def test_choose_transport_takes_the_first_that_passed_on_every_rank(codes, expect, asked):
    # two ranks emulated in one process: the "collective" is the minimum over both ranks' results for that candidate
    for rank in range(2):
        seen = []

        def probe(t):
            seen.append(t)
            return codes[t][rank]

        def all_min(ok):
            t = seen[-1]
            assert ok == (1 if codes[t][rank] == 0 else 0)
            return min(1 if c == 0 else 0 for c in codes[t])
        lines = []
        assert P.choose_transport(["ipc", "rccl", "shm"], probe, all_min, lines.append) == expect
        assert seen == asked and len(lines) == len(asked)   # every rank asks the same candidates in the same order


def _probe_ranks(transport, world, name, n=32):
    import multiprocessing.pool
    env_old = os.environ.get("FASP_HIP_ALLOW_DEVICE_WRAP")
    os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"] = "1"
    try:
        with multiprocessing.pool.ThreadPool(world) as tp:
            return tp.map(lambda r: P.run_child(transport, r, world, 0, name, n=n, timeout_s=300), range(world))
    finally:
        if env_old is None:
            del os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"]
        else:
            os.environ["FASP_HIP_ALLOW_DEVICE_WRAP"] = env_old


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["ipc", "shm"])
def test_probe_passes_over_transports_that_work_here(gpu, transport):
    assert _probe_ranks(transport, 2, f"fasp_tprobe_{os.getpid()}_{transport}") == [0, 0]


@pytest.mark.gpu
def test_probe_reports_a_transport_that_cannot_start(gpu):
    # an unknown transport: every rank's child says so with exit code 2, nobody waits for anybody
    assert _probe_ranks("nosuch", 2, f"fasp_tprobe_{os.getpid()}_x") == [2, 2]


def _choose_worker(rank, world, port, codes, q):
    import datetime

    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    def all_min(ok):
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0])
    asked = []
    got = P.choose_transport(["ipc", "rccl", "shm"], lambda t: (asked.append(t), codes[t][rank])[1], all_min)
    q.put((rank, got, asked))
    dist.destroy_process_group()


def test_choose_transport_over_a_real_collective():
    """Two processes, gloo, world_size 2 (the rendezvous of bench_dist.py): peer windows pass on rank 0 and time out on rank 1, RCCL
    cannot start on rank 0 -> both ranks end up on shared memory, having asked the same three candidates in the same order."""
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    codes = {"ipc": [0, 124], "rccl": [2, 0], "shm": [0, 0]}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_choose_worker, args=(r, 2, port, codes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert [r[1] for r in res] == ["shm", "shm"]
    assert all(r[2] == ["ipc", "rccl", "shm"] for r in res)
