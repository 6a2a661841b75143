"""The driver's literal scaling command (the contract of bench.py): `python bench.py --gpus N --steps K --warmup W` without a
launcher -- bench.py starts `python -m torch.distributed.run` as a child, the ranks rendezvous over gloo on 127.0.0.1, probe the
transports in child processes of their own (peer windows, RCCL, shared memory: faspsolver_amd/comm_probe.py), row-partition the
problem, solve, and rank 0 prints ONE JSON line.  On a box with one GPU the ranks share it (validation set-up, labelled in the
line by the transport's name); on the 8-GPU node the same command is the measurement.  SURVEY.md section 8(e).

These tests are the first place the driver's environment executes that path: rc 0, one JSON line, n_gpus as asked, the
transport named, the single-GPU iteration count, the reference's own figures where they are pinned.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(n, gpus, comm, steps=2, warmup=1, timeout=900):
    env = dict(os.environ)
    env["BENCH_N"] = str(n)
    env["BENCH_COMM"] = comm
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
                        "--no-cpu-baseline"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["auto", "rccl"])
def test_bench_scaling_command_two_ranks(comm):
    """BENCH_N=64 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline.  comm = auto: the transport is whatever the probes
    find working on every rank; comm = rccl on a box whose ranks share one GPU: the communicator cannot be created there and the
    run falls back to the peer-window transport TOGETHER (every rank, cleanly) instead of hanging or failing."""
    p, lines = _run_bench(64, 2, comm)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["unit"] == "DOF/s" and out["value"] > 0 and out["scaling"] == "strong" and out["dtype"] == "f64"
    assert out["comm"]["transport"] in ("ipc", "rccl", "shm"), out["comm"]
    assert out["iterations"] == 9                      # P7(64): the single-GPU count = the reference's (tests/golden/p7_scale.npz)
    assert abs(out["relres"] - 3.0782769324e-09) <= 1e-10
    assert out["max_abs_error_vs_exact"] < 5e-4        # second-order discretisation error at h = 1/65
    assert out["comm"]["per_solve"]["halo_exchanges"] > 0 and out["comm"]["per_solve"]["allreduces"] > 0
    pr = out.get("parity_reference")
    assert pr is None or pr["ok"], pr
    assert "roofline" in out and out["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
def test_bench_scaling_command_at_the_metric_size_two_ranks():
    """The same command at BASELINE.json's own size, P7(256) on two ranks: 14 iterations and the reference's residual."""
    if os.environ.get("FASP_TEST_SCALING_256", "1") == "0":
        pytest.skip("FASP_TEST_SCALING_256=0")
    p, lines = _run_bench(256, 2, "auto", steps=2, warmup=1, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["iterations"] == 14
    pr = out["parity_reference"]
    assert pr["ok"] and pr["iters_reference"] == 14, pr


@pytest.mark.gpu
def test_ranks_on_distinct_devices_over_rccl_when_several_gpus_are_visible():
    """with >= 2 devices visible, `python bench.py --gpus 2` at the metric's size must (i) put every rank on its own
    device (PCI bus ids in comm.devices all different), (ii) carry the halos and all-reduces over RCCL (comm.transport == "rccl",
    comm.rccl_ranks == 2) and (iii) hold the two-rank pins of P7(256): 14 iterations, the reference's residual.  On a box with ONE visible
    device -- every box this repository has run on so far -- it SKIPS, and the reason says that no run with ranks on different GPUs has
    happened yet."""
    import faspsolver_amd as fa
    ndev = fa.lib().fasp_hip_device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: ranks on DISTINCT devices over RCCL need at least two -- this path has never run on this box "
                    "(the shared-device validation runs are the tests above)")
    p, lines = _run_bench(256, 2, "rccl", steps=2, warmup=1, timeout=1500)   # (BENCH_COMM=rccl: north_star's transport, no probing)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(lines[0])
    c = out["comm"]
    assert c["ranks_on_distinct_devices"] and len(set(c["devices"])) == 2, c["devices"]
    assert c["transport"] == "rccl" and c["rccl_ranks"] == 2, c
    assert out["iterations"] == 14 and out["parity_reference"]["ok"], out["parity_reference"]
