"""Transport probe: ONE rank of a small partitioned solve over the named transport, in a process of its own.

    python -m faspsolver_amd.comm_probe <rccl|ipc|shm> <rank> <world> <device> <name> [n]

bench_dist.py starts one of these per rank BEFORE it touches the GPU itself and reads the exit codes: a transport that
cannot be set up on this node -- or faults, or times out, or gives a solve that differs from the same rank's unpartitioned
solve -- ends the child with a non-zero code and costs the bench nothing but the seconds of the probe; the ranks then agree
on the first transport that passed on every one of them (choose_transport).  The development boxes have one GPU, so
the first run on a multi-GPU node is also the first time RCCL carries more than one rank: the probe is what keeps that run
from ending without a number.

What a passing probe has exercised: the communicator's start-up (unique id / shared segment / hipIpc handles), halo
exchanges on every partitioned level, the batched all-reduces of PCG, the all-gathers in front of the replicated levels,
and the finalize -- through the library's own solve, compared with the unpartitioned solve of the same system computed
in this process first (equal iteration count, final relative residuals within 1e-5 of each other -- the sums run in another
order --, own rows of x within 1e-8 max|x|).

Exit codes: 0 passed, 1 a check failed, 2 the transport could not be initialised, 3 exception.
"""
import os
import sys
import time


def visible_gpu_count():
    """Number of GPUs the HIP runtime of THIS environment offers, asked in a CHILD process (hipGetDeviceCount through the library):
    the caller -- a rank that is about to start the probe children and must not have initialised the GPU before it does -- never touches
    the runtime itself, and the answer is the runtime's own (the KFD topology and the render nodes list every GPU of the node even when
    the container may use one of them: counting there sent rank 1 of a two-rank run on a one-GPU box to a device that does not exist).
    0 when the child fails (no GPU, no library)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); import faspsolver_amd as fa; print('NDEV', fa.lib().fasp_hip_device_count())" % root
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120).stdout
        for line in out.splitlines():
            if line.startswith("NDEV"):
                return max(0, int(line.split()[1]))
    except Exception:
        pass
    return 0


def choose_transport(candidates, probe, all_min, log=lambda s: None):
    """The first transport of `candidates` whose probe passed on EVERY rank (None if there is none).
    probe(name) -> this rank's exit code; all_min(ok) -> the minimum of `ok` over the ranks (a collective: every rank
    calls it once per candidate, in the same order)."""
    for t in candidates:
        rc = probe(t)
        ok = all_min(1 if rc == 0 else 0)
        log(f"transport probe {t}: rc {rc} here, {'passed' if ok else 'FAILED'} over the ranks")
        if ok:
            return t
    return None


def candidates(ndev, world):
    """Order of preference: peer windows (one kernel per exchange), RCCL (the stock transport), host-staged shared memory
    (works wherever the ranks share a node).  RCCL needs a device per rank."""
    return ["ipc", "rccl", "shm"] if ndev >= world else ["ipc", "shm"]


def run_child(transport, rank, world, dev, name, n=48, timeout_s=240, out=None):
    """Start the probe as a child process and wait for it: its exit code, or 124 on a timeout (the child is killed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("FASP_HIP_SHM_TIMEOUT_S", "30")
    env["OMP_NUM_THREADS"] = env.get("OMP_NUM_THREADS", "4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "faspsolver_amd.comm_probe", transport, str(rank), str(world), str(dev), name, str(n)]
    try:
        p = subprocess.run(cmd, env=env, stdout=out if out is not None else sys.stderr, stderr=subprocess.STDOUT, timeout=timeout_s, cwd=root)
        return p.returncode
    except subprocess.TimeoutExpired:
        return 124


def _main(argv):
    transport, rank, world, dev, name = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
    n = int(argv[5]) if len(argv) > 5 else 48
    # levels of at least n^3 / 16 rows are partitioned: the finest two or three of P7(48)
    os.environ.setdefault("FASP_HIP_DIST_MIN_ROWS", str(max(64, n * n * n // 16)))
    import ctypes as C
    import numpy as np
    import faspsolver_amd as fa
    from faspsolver_amd import _types as T
    L = fa.lib()
    if L.fasp_hip_set_device(dev) != 0:
        print(f"[probe {transport} rank {rank}] no device {dev}", flush=True)
        return 2
    ia, ja, a, f, ue = fa.poisson7pt(n)
    itp = fa.param_solver_init(); itp.tol = 1e-8
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    # the unpartitioned solve first (no communicator yet): what every rank must reproduce
    H = fa.AMG(ia, ja, a, amgp)
    st0, x0, hist0, stats0 = H.solve(f, itp)
    H.close()
    if transport == "rccl":
        idfile = f"/dev/shm/{name}.ncclid"
        if rank == 0:
            buf = C.create_string_buffer(128)
            if L.fasp_hip_comm_unique_id(buf) != 0:
                return 2
            with open(idfile + ".tmp", "wb") as fh:
                fh.write(buf.raw)
            os.replace(idfile + ".tmp", idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                print(f"[probe rccl rank {rank}] no unique id from rank 0", flush=True)
                return 2
            time.sleep(0.02)
        st = L.fasp_hip_comm_init(rank, world, open(idfile, "rb").read())
    elif transport == "ipc":
        st = L.fasp_hip_comm_init_ipc(rank, world, name.encode())
    elif transport == "shm":
        st = L.fasp_hip_comm_init_shm(rank, world, name.encode())
    else:
        print(f"[probe] unknown transport {transport}", flush=True)
        return 2
    if st != 0:
        print(f"[probe {transport} rank {rank}] communicator init -> {st}", flush=True)
        return 2
    H = fa.AMG(ia, ja, a, amgp)
    info = H.dist_info(0)
    rc = 0
    for rep in range(2):   # twice: the second solve runs over warm windows / connections
        st1, x1, hist1, stats1 = H.solve(f, itp)
        lo, hi = info["row0"], info["row0"] + info["nloc"]
        dx = float(np.max(np.abs(x1[lo:hi] - x0[lo:hi]))) if hi > lo else 0.0
        ok = (info["replicated"] == 0 and st1 == st0 and st1 > 0
              and abs(stats1.relres - stats0.relres) <= 1e-5 * stats0.relres
              and dx <= 1e-8 * float(np.max(np.abs(x0))))
        if not ok:
            print(f"[probe {transport} rank {rank}] MISMATCH: iterations {st1} vs {st0}, relres {stats1.relres:.10e} vs {stats0.relres:.10e}, "
                  f"max |dx| own rows {dx:.3e}, level 0 replicated {info['replicated']}", flush=True)
            rc = 1
    H.close()
    L.fasp_hip_comm_finalize()
    if transport == "rccl" and rank == 0:
        try:
            os.remove(f"/dev/shm/{name}.ncclid")
        except OSError:
            pass
    if rc == 0:
        print(f"[probe {transport} rank {rank}/{world} dev {dev}] passed: P7({n}), {st1} iterations, rows {info['nloc']} (+{info['nghost']} ghosts)", flush=True)
    return rc


if __name__ == "__main__":
    try:
        code = _main(sys.argv[1:])
    except BaseException as e:   # noqa: BLE001 -- the exit code is the report
        import traceback
        traceback.print_exc()
        code = 3
    sys.stdout.flush()
    os._exit(code)   # (no interpreter tear-down after a failed collective: peers may be gone)
