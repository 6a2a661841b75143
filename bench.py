#!/usr/bin/env python3
"""bench.py -- AMG-PCG solve throughput on the headline workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 3-D 7-point FD Poisson on the unit cube, 256^3
interior points (16 777 216 DOF, 117 047 296 nnz), rhs from the reference's generator,
x0 = 0, PCG rtol 1e-8, classical RS-AMG defaults with SMOOTHER_JACOBI, relaxation 0.6667,
V(1,1).  One "step" = one complete fasp_solver_dcsr_krylov_amg-style
Krylov solve (all PCG iterations, every V-cycle, the coarse-level safe CG) on the
resident hierarchy, with b and x already in HBM.  The host-side AMG setup and the
upload happen once, before the warm-up, and are reported separately.

value    = DOF / s = (rows x K) / (time of K solves), whole job (all ranks)
roofline = the level-0 SpMV kernel the solve runs (t = A p fused with the (t,p) partial
           sums): bytes that kernel HAS TO MOVE per launch (its stored matrix form + x once
           + y once) / mean launch time measured with HIP events on the launch stream inside
           the timed solves; frac = that / 8 TB/s, never above 1.  traffic = memory-side bytes
           per launch from the rocprofv3 PMC passes of this same command (tools/profile.sh ->
           profiles/r05_rocprof/traffic.json, falling back to the previous round's: 2 x FETCH_SIZE +
           WRITE_SIZE, separate passes).
roofline_plain_csr = the same operator through the plain-CSR kernel (lossless coding switched
           off for these launches): SURVEY 8(d)'s algorithmic bytes 12 nnz + 4 (m+1) + 8 m + 8 m
           are exactly what this kernel moves.  This is north_star's "SpMV >= 60 % of the HBM roofline".
ceilings = measured 16-byte-per-lane read / copy / triad rates of this device (1 GiB buffers).
variable_coefficient = a second full solve: -div(kappa grad u) with a smooth kappa of contrast 9 on
           the same grid; no two rows repeat, so every level runs the plain-CSR kernels.
other_configs = configs 3 and 5 of BASELINE.json at full size and the reference's DEFAULT smoother (Gauss-Seidel in C/F
           order, the reference's sequential sweep reproduced: csrc/seq_split.hip.h) at 128^3 and 256^3 -- one warm-up solve, then
           the mean of three timed solves each, with the reference's iteration count beside the measured one.
cpu_baseline = the oracle (oracle/liboracle.so, a plain-C restatement of the reference's
           serial algorithm with OpenMP row loops) on the node's host cores, same problem, same
           hierarchy, a bounded number of PCG iterations scaled to the full solve; once on a team
           (threads pinned one per physical core over both sockets, pages first touched by the thread
           that reads them; the team size that is fastest on this host, of all physical cores and
           halves of that) and once on one thread.

N > 1: `python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child (unless the driver already did: WORLD_SIZE set); the matrix is
row-partitioned, halos and dot products go over the first transport whose probe passes on every
rank -- peer windows over hipIpc-mapped device memory, then RCCL, then host-staged shared memory
(BENCH_COMM=auto; faspsolver_amd/comm_probe.py, csrc/comm.cpp, comm_ipc.hip, dist_plan.cpp; bench_dist.py).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

# torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the host-side AMG setup is
# OpenMP code, so give each rank its share of the node's cores before any OpenMP runtime
# is initialised (this must precede the numpy / torch imports).
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("BENCH_KEEP_OMP") is None:
    _cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(32, _cores // int(os.environ["WORLD_SIZE"]))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import faspsolver_amd as fa  # noqa: E402
from faspsolver_amd import _types as T  # noqa: E402

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# PMC traffic summaries of THIS command, one per workload (tools/profile.sh <tag> <workload> -> tools/summarize_prof.py);
# a figure is attached to a roofline entry only when kernel name, workload and grid size all match the run
TRAFFIC_JSONS = {"constant": [os.path.join("profiles", r + "_rocprof", "traffic.json") for r in ("r06", "r05", "r04")],
                 "variable": [os.path.join("profiles", r + "_rocprof_var", "traffic.json") for r in ("r06", "r05", "r04")]}

# kernel family codes of fasp_hip_amg_kernel_info -> (rocprofv3 kernel name of the OP_MXV_DOT instantiation, description)
KERNELS = {0: ("k_csr_rows", "k_csr_rows<L, OP_MXV_DOT> (sub-wavefront per row, plain CSR)"),
           2: ("k_csr_wstream<7, 64, 512>", "k_csr_wstream<OP_MXV_DOT,64,512> (wave-level stream, plain CSR)"),
           4: ("k_csr_dict8<7", "k_csr_dict8<OP_MXV_DOT> (one byte per entry: (column offset, value) dictionary)"),
           5: ("k_csr_rowpat<7", "k_csr_rowpat<OP_MXV_DOT> (one 16-bit row-pattern id per row)"),
           6: ("k_csr_rowpat4<7>", "k_csr_rowpat4<OP_MXV_DOT> (16-bit row-pattern ids, scalar-pattern sweep, rows off the wave's pattern through an in-sweep LDS queue)"),
           7: ("k_csr_lstream<7, 512>", "k_csr_lstream<OP_MXV_DOT,512> (16-byte staged stream, lane = row, plain CSR)"),
           8: ("k_csr_wstream2<7>", "k_csr_wstream2<OP_MXV_DOT> (16-byte staged, prefetched wave stream, plain CSR)"),
           9: ("k_csr_rowpat5<7>", "k_csr_rowpat5<OP_MXV_DOT> (16-bit row-pattern ids, pair-of-patterns sweep, rectangular)"),
           10: ("k_csr_xtile<7>", "k_csr_xtile<OP_MXV_DOT> (tile's distinct x entries in LDS, 16-bit column positions; lossless)")}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spmv_bytes(m, ncol, nnz):
    return 12 * nnz + 4 * (m + 1) + 8 * ncol + 8 * m


def workload_params():
    itp = fa.param_solver_init()
    itp.tol = 1e-8
    itp.maxit = 500
    itp.print_level = 0
    amgp = fa.param_amg_init()
    amgp.smoother = T.SMOOTHER_JACOBI
    amgp.relaxation = 0.6667
    return itp, amgp


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def physical_cores():
    """One logical CPU per physical core of the CPUs this process may run on, in CPU order (sockets one after the other)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return list(range(os.cpu_count() or 1))
    out = []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
            first = int(sib.replace("-", ",").split(",")[0])
        except Exception:
            first = c
        if first == c or first not in allowed:
            out.append(c)
    return out or allowed


def baseline_threads():
    """Threads of the all-core CPU baseline: every physical core (BENCH_CPU_THREADS caps it)."""
    cap = int(os.environ.get("BENCH_CPU_THREADS", "0"))
    n = len(physical_cores())
    return max(1, min(n, cap) if cap > 0 else n)


def cpu_baseline(H, ia, ja, a, f, iters_gpu, hist_gpu, budget_s, threads):
    """Time the oracle on the host cores on a bounded sample of the same solve.  threads > 1: the team is pinned, one thread
    per physical core spread over the sockets, and the hierarchy is copied so that every page is first touched by the
    thread that reads it (oracle/fasp_oracle.c, timing mode) -- what an OpenMP code does on a NUMA host.
    threads = a list: one PCG iteration is timed at each count and the sample runs at the fastest (on a host that other
    jobs share, the team that finishes first is not the largest)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs
    O = _libs.oracle()
    O.orc_get_threads.restype = C.c_int
    # the oracle and libfasp_hip share one OpenMP runtime: orc_set_threads() moves the calling thread's nthreads ICV,
    # which the host AMG setup reads afterwards (round 2's variable-coefficient setup ran on ONE thread behind the
    # one-thread baseline: 24.7 s instead of 10 s) -- restored on the way out
    omp_before = int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores()
    nl = H.num_levels
    A, _k = T.as_csr(ia, ja, a)
    bv, _f = T.as_vec(f)

    class Session:
        def __init__(self, nthreads):
            self.n = nthreads
            O.orc_set_threads(nthreads)
            self.buf = C.create_string_buffer(O.orc_sizeof_amg())
            O.orc_amg_borrow_begin(self.buf, nl)
            self.keep = []
            for l in range(nl):
                vA = T.dCSRmat(); fa.lib().fasp_hip_amg_get_matrix(H.h, l, 0, C.byref(vA))
                if l < nl - 1:
                    vP = T.dCSRmat(); vR = T.dCSRmat(); cf = T.ivector()
                    fa.lib().fasp_hip_amg_get_matrix(H.h, l, 1, C.byref(vP))
                    fa.lib().fasp_hip_amg_get_matrix(H.h, l, 2, C.byref(vR))
                    fa.lib().fasp_hip_amg_get_cfmark(H.h, l, C.byref(cf))
                    O.orc_amg_borrow_level(self.buf, l, C.byref(vA), C.byref(vP), C.byref(vR), cf.val)
                    self.keep += [vA, vP, vR, cf]
                else:
                    O.orc_amg_borrow_level(self.buf, l, C.byref(vA), None, None, None)
                    self.keep += [vA]
            O.orc_amg_borrow_end(self.buf)
            self.pinned = False
            if nthreads > 1:
                cores = physical_cores()
                # thread t -> the t-th of `nthreads` cores spread evenly over the list (both sockets at any thread count)
                pick = [cores[(t * len(cores)) // nthreads] for t in range(nthreads)] if nthreads <= len(cores) else cores
                arr = (C.c_int * len(pick))(*pick)
                self.pinned = O.orc_pin_threads(arr, len(pick)) == 0
                O.orc_amg_place(self.buf)

        def run(self, maxit):
            itp, amgp = workload_params()
            itp.maxit = maxit
            x = np.zeros(len(f))
            xv, x = T.as_vec(x)
            hist = np.zeros(600); nh = C.c_int(0); rr = C.c_double(0)
            t0 = time.perf_counter()
            st = O.orc_solve_with_hierarchy(self.buf, C.byref(A), C.byref(bv), C.byref(xv), C.byref(itp),
                                            C.byref(amgp), T.dp(hist), 600, C.byref(nh), C.byref(rr))
            return time.perf_counter() - t0, st, hist[:nh.value].copy(), rr.value

        def close(self):
            O.orc_amg_borrow_free(self.buf)
            if self.pinned:
                O.orc_unpin_threads()

    tried = None
    if isinstance(threads, (list, tuple)):
        tried = {}
        for n in threads:
            S = Session(n)
            S.run(1)                      # (first touch of the work vectors)
            tried[n] = S.run(1)[0]
            S.close()
        threads = min(tried, key=tried.get)
    S = Session(threads)
    run = S.run
    if threads > 1:
        run(1)
    t1, st1, h1, _ = run(1)  # 1 iteration (+ the initial preconditioner apply): sizes the sample
    per_it = max(t1 / 2.0, 1e-6)
    k = int(max(1, min(iters_gpu, budget_s / per_it)))
    if k >= iters_gpu:
        tk, stk, hk, rrk = run(500)
        sample = f"full solve, {stk} PCG iterations"
        t_full = tk
        its_cpu = stk
    elif k == 1:
        tk, hk = t1, h1
        t_full = tk * (iters_gpu + 1.0) / 2.0
        sample = (f"1 of {iters_gpu} PCG iterations of the same solve ({tk:.2f} s), scaled by ({iters_gpu}+1)/2")
        its_cpu = None
        rrk = None
    else:
        tk, stk, hk, rrk = run(k)
        # k iterations contain k+1 preconditioner applies; the full solve iters+1
        t_full = tk * (iters_gpu + 1.0) / (k + 1.0)
        sample = (f"{k} of {iters_gpu} PCG iterations of the same solve "
                  f"({tk:.2f} s), scaled by ({iters_gpu}+1)/({k}+1)")
        its_cpu = None
        rrk = None
    ncmp = min(len(hk) - 1, len(hist_gpu) - 1)
    hist_dev = float(np.max(np.abs(hk[:ncmp] - hist_gpu[:ncmp]) / hk[:ncmp])) if ncmp > 0 else None
    pinned = S.pinned
    S.close()
    O.orc_set_threads(omp_before)
    out = {"value": len(f) / t_full, "unit": "DOF/s", "cores": threads, "kind": "port",
           "sample": sample, "seconds_full_solve_est": t_full, "cpu_model": cpu_model(),
           "host_cores_visible": host_cores(),
           "placement": ("threads pinned one per physical core, pages first touched by the reading thread" if pinned else
                         "one thread" if threads == 1 else "threads not pinned")}
    if tried is not None:
        out["seconds_per_iteration_by_threads"] = {str(n): round(v, 4) for n, v in tried.items()}
    return out, its_cpu, rrk, hist_dev


def baseline_candidates():
    """Thread counts the all-core baseline tries: every physical core, and halves of it down to 8."""
    top = baseline_threads()
    c, n = [], top
    while n >= 8:
        c.append(n); n //= 2
    return c or [top]


def pmc_traffic(kernel_name, workload, n):
    """(bytes per launch, source file) of a kernel from the committed PMC summary of this command for THIS workload
    and grid size; (None, None) when there is no pass of exactly this kernel instantiation on this workload."""
    for rel in TRAFFIC_JSONS.get(workload, []):   # this round's summary, else the previous round's
        try:
            tj = json.load(open(os.path.join(ROOT, rel)))
            if tj.get("workload") != workload or int(tj.get("n", -1)) != int(n):
                continue
            rec = tj.get("kernels", {}).get(kernel_name)   # exact instantiation name, e.g. "k_csr_lstream<7, 512>"
            if rec is not None:
                return float(rec["bytes_per_launch"]), rel
        except Exception:
            continue
    return None, None


def solve_wide_traffic(workload, n):
    """(HBM-side bytes per PCG iteration of the whole solve, source file) from the committed PMC summary of this command
    (tools/summarize_prof.py: 2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of the timed solves / iterations), or None."""
    for rel in TRAFFIC_JSONS.get(workload, []):
        try:
            tj = json.load(open(os.path.join(ROOT, rel)))
            if tj.get("workload") != workload or int(tj.get("n", -1)) != int(n):
                continue
            sw = tj.get("solve_wide")
            if sw and sw.get("bytes_per_iteration"):
                return float(sw["bytes_per_iteration"]), rel
        except Exception:
            continue
    return None


def roofline_entry(kind, moved_bytes, kernel_ms, launches, note=None, workload=None, n=None, lanes=None):
    name, desc = KERNELS.get(kind, (str(kind), str(kind)))
    if kind == 0 and lanes:
        name = f"k_csr_rows<{lanes}, 7>"
    achieved = moved_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, traffic_source = pmc_traffic(name, workload, n) if workload else (None, None)
    out = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
           "kernel": "level-0 t = A p fused with (t,p): " + desc,
           "bytes_per_launch": moved_bytes, "ms_per_launch": kernel_ms, "launches_timed": launches,
           "traffic_source": traffic_source,
           "traffic_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic and kernel_ms > 0 else None,
           "traffic_over_bytes": (traffic / moved_bytes) if traffic else None}
    if note:
        out["note"] = note
    return out


def spawn_ranks(args):
    """--gpus N without a launcher: start the ranks as a child job (never re-exec this process)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    log("bench.py: launching", " ".join(cmd))
    env = dict(os.environ)
    env["BENCH_N"] = str(args.n)   # (torch.distributed.run's parser trips over a script option called --n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def variable_leg(n, system, itp, amgp, timed_solves):
    """Variable coefficients: -div(kappa grad u) on the same grid.  No two rows repeat, so no lossless coding applies
    and every level runs the plain-CSR kernels (what an application matrix sees)."""
    ia, ja, a, f, ue = system
    m = len(f)
    try:
        _ia, _ja, a2, f2 = fa.poisson7pt_var(n, (ia, ja, a, f, ue))
        t0 = time.perf_counter()
        H2 = fa.AMG(ia, ja, a2, amgp)
        ts2 = time.perf_counter() - t0
        H2.set_rhs(f2)
        el2, st2, hist2, stats2, kms2 = timed_solves(H2, 1, 3)
        k2, mb2 = H2.kernel_info(0, 0)
        res = {
            "workload": f"-div(kappa grad u) on the grid of P7({n}), kappa = 1 + 0.8 sin(2 pi x) sin(2 pi y) sin(2 pi z) "
                        "(contrast 9, no repeated rows); same solver parameters; 3 timed solves",
            "value": m * 3 / el2, "unit": "DOF/s", "ms_per_step": 1e3 * el2 / 3, "iterations": int(st2),
            "relres": stats2.relres, "setup_seconds": ts2, "levels": H2.num_levels,
            "roofline": roofline_entry(k2, mb2 + 16.0 * m, kms2, int(stats2.spmv_launches) * 3, workload="variable", n=n)}
        log(f"variable-coefficient solve: {st2} iterations, {1e3*el2/3:.2f} ms/solve, setup {ts2:.1f} s, level-0 kernel family {k2}, "
            f"SpMV {kms2*1e3:.1f} us")
        H2.close()
        return res
    except Exception as e:
        log(f"variable-coefficient solve failed: {e!r}")
        return None


def device_state():
    """Clocks and temperatures of GPU 0 read from sysfs (memory clock, fabric clock, junction / HBM temperature, package power) --
    printed next to the measured ceilings: two boxes of the pool whose triad ceilings agree to 1 % have differed by 14 % on the
    plain-CSR level-0 kernel; this is what can be read without privileges to tell them apart.
    No child process: rocm-smi is an `env python3` script, and under `rocprofv3 --pmc` (tools/profile.sh) a child
    inherits the profiler's preload -- a GPU-initialised process replacing its program, which this pool must never see."""
    import glob
    try:
        devs = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002" and os.path.exists(os.path.join(d, "pp_dpm_mclk")):
                    devs.append(d)
            except OSError:
                pass
        if not devs:
            return {"unavailable": "no amdgpu device with pp_dpm_mclk under /sys/class/drm"}
        d = devs[0]

        def active(name):   # the line of a pp_dpm_* table that carries the '*'
            try:
                for ln in open(os.path.join(d, name)).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        return ln.split(":", 1)[1].replace("*", "").strip()
            except OSError:
                pass
            return ""

        res = {"mclk": active("pp_dpm_mclk"), "fclk": active("pp_dpm_fclk"), "sclk_idle": active("pp_dpm_sclk"), "source": "sysfs"}
        for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for lab in glob.glob(os.path.join(h, "temp*_label")):
                try:
                    name = open(lab).read().strip()
                    val = int(open(lab.replace("_label", "_input")).read()) / 1000.0
                except (OSError, ValueError):
                    continue
                if name == "junction":
                    res["temp_junction_C"] = val
                elif name == "mem":
                    res["temp_hbm_C"] = val
            for pw in ("power1_average", "power1_input"):
                try:
                    res["package_power_W"] = int(open(os.path.join(h, pw)).read()) / 1e6
                    break
                except (OSError, ValueError):
                    continue
        return res
    except Exception as e:   # (no sysfs access: the line just lacks the field)
        return {"unavailable": repr(e)}


TIMED_SOLVES = 3
GS256_ITERS_REF = SOR256_ITERS_REF = None   # filled in from tests/golden/p7_sweeps_256.npz when that fixture is present
try:
    GS256_ITERS_REF = int(np.load(os.path.join(ROOT, "tests", "golden", "p7_sweeps_256.npz"))["gscf_iters"])
    SOR256_ITERS_REF = int(np.load(os.path.join(ROOT, "tests", "golden", "p7_sweeps_256.npz"))["sor11_iters"])
except Exception:
    pass


def other_configs_leg():
    """The other single-GPU configurations of BASELINE.json at full size, and the reference's DEFAULT smoother: one timed solve
    each on a resident hierarchy (second solve: the first one builds lazily what it needs).  Reported beside the headline,
    never instead of it; iteration counts are the compiled reference's (tests/golden/configs_full.npz, p7_sweeps.npz pin
    them in the test suite)."""
    from faspsolver_amd import _types as T
    res = {}
    try:   # config 3: P7(128) (x) B3, UA-AMG (VMB) + block Jacobi + VGMRES(30)
        ia, ja, a, f0, ue = fa.poisson7pt(128)
        B3 = np.array([[4.0, 1.0, 0.0], [1.0, 3.0, 1.0], [0.0, 1.0, 2.0]])
        val = (a[:, None, None] * B3[None, :, :]).reshape(-1)
        f = np.random.default_rng(1).standard_normal((len(ia) - 1) * 3)
        itp, amgp = fa.param_solver_init(), fa.param_amg_init()
        amgp.AMG_type = T.UA_AMG; amgp.aggregation_type = 2; amgp.smoother = T.SMOOTHER_JACOBI; amgp.cycle_type = 1
        itp.tol = 1e-8; itp.itsolver_type = 5; itp.restart = 30
        t0 = time.perf_counter()
        G = fa.BSRAMG(ia, ja, val, 3, amgp)
        ts = time.perf_counter() - t0
        secs = []
        for rep in range(1 + TIMED_SOLVES):
            st, x, hist, stats = G.solve(f, itp)
            secs.append(stats.solve_seconds)
        sec = float(np.mean(secs[1:]))
        res["config3"] = {"workload": "P7(128) (x) B3 (6.3 M DOF, 14.6 M blocks), UA-AMG (VMB) + block Jacobi + VGMRES(30), rtol 1e-8",
                          "ms_per_solve": sec * 1e3, "timed_solves": TIMED_SOLVES, "iterations": int(st), "iterations_reference": 66,
                          "relres": stats.relres, "DOF_per_s": len(f) / sec, "setup_seconds": ts}
        G.free()
        log(f"config 3: {st} iterations, {sec*1e3:.1f} ms")
    except Exception as e:
        log(f"config 3 leg failed: {e!r}")
    try:   # config 5: anisotropic 27-point operator, SA-AMG + W-cycle + VFGMRES(30)
        ia, ja, a, f = fa.aniso27pt(123)
        itp, amgp = fa.param_solver_init(), fa.param_amg_init()
        itp.tol = 1e-8; itp.itsolver_type = 6; itp.restart = 30
        amgp.AMG_type = T.SA_AMG; amgp.cycle_type = T.W_CYCLE; amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
        t0 = time.perf_counter()
        H = fa.AMG(ia, ja, a, amgp)
        ts = time.perf_counter() - t0
        H.set_rhs(f)
        secs = []
        for rep in range(1 + TIMED_SOLVES):
            st, hist, stats = H.solve_resident(itp)
            secs.append(stats.solve_seconds)
        sec = float(np.mean(secs[1:]))
        res["config5"] = {"workload": "Q1 27-point, anisotropy (1, 1, 0.01), n = 123 (1.86 M rows, 49.4 M nnz), SA-AMG + W-cycle + VFGMRES(30), w-Jacobi, rtol 1e-8",
                          "ms_per_solve": sec * 1e3, "timed_solves": TIMED_SOLVES, "iterations": int(st), "iterations_reference": 89,
                          "relres": stats.relres, "DOF_per_s": len(f) / sec, "setup_seconds": ts}
        H.close()
        log(f"config 5: {st} iterations, {sec*1e3:.1f} ms")
    except Exception as e:
        log(f"config 5 leg failed: {e!r}")
    # the reference's default smoother (Gauss-Seidel, C/F order) in the parity mode, at 128^3 and at the size of the metric;
    # reference iteration counts: tests/golden/p7_sweeps_128.npz / p7_sweeps_256.npz (the compiled reference's own runs)
    # ... and SOR(1.1) in natural order (north_star names SOR) at the size of the metric: 10 iterations in the compiled reference
    for n, its_ref, key, what, mod in ((128, 8, "gs_defaults_128", "fasp_param_amg_init defaults (GS smoother in C/F order, the reference's sequential sweep)", None),
                                       (256, GS256_ITERS_REF, "gs_defaults_256", "fasp_param_amg_init defaults (GS smoother in C/F order, the reference's sequential sweep)", None),
                                       (256, SOR256_ITERS_REF, "sor11_256", "SOR(1.1) smoother in natural order (the reference's sequential sweep)", "sor")):
        try:
            ia, ja, a, f, ue = fa.poisson7pt(n)
            itp, amgp = fa.param_solver_init(), fa.param_amg_init()
            itp.tol = 1e-8
            if mod == "sor":
                amgp.smoother = T.SMOOTHER_SOR; amgp.relaxation = 1.1; amgp.smooth_order = 0
            t0 = time.perf_counter()
            H = fa.AMG(ia, ja, a, amgp)
            ts = time.perf_counter() - t0
            H.set_rhs(f)
            secs = []
            for rep in range(1 + TIMED_SOLVES):   # (the first solve also builds the sweep schedules of every level)
                st, hist, stats = H.solve_resident(itp)
                secs.append(stats.solve_seconds)
            sec = float(np.mean(secs[1:]))
            res[key] = {"workload": f"P7({n}), {what} + PCG, rtol 1e-8",
                                       "ms_per_solve": sec * 1e3, "timed_solves": TIMED_SOLVES, "first_solve_ms": secs[0] * 1e3, "iterations": int(st),
                                       "iterations_reference": its_ref, "relres": stats.relres, "DOF_per_s": len(f) / sec, "setup_seconds": ts}
            H.close()
            log(f"{key}: {st} iterations, {sec*1e3:.1f} ms (first solve, with the schedules: {secs[0]*1e3:.0f} ms)")
        except Exception as e:
            log(f"{key} leg failed: {e!r}")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", "256")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variable", action="store_true", help="skip the variable-coefficient second solve")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configurations (configs 3 and 5 at full size, GS defaults at 128^3 and 256^3)")
    ap.add_argument("--no-plain", action="store_true", help="skip the extra solves with the coding switched off (profiling runs)")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the read / copy / triad ceiling kernels (profiling runs)")
    ap.add_argument("--only-variable", action="store_true",
                    help="profiling runs: only the variable-coefficient solve (tools/profile.sh <tag> variable)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # before anything touches the GPU in this process
    if world != args.gpus and world > 1:
        log(f"WORLD_SIZE {world} != --gpus {args.gpus}; using WORLD_SIZE")
    if world > 1:
        import bench_dist
        return bench_dist.main(args, rank, world, local_rank)

    L = fa.lib()
    if not fa.available():
        log("bench.py: no HIP device; libfasp_hip has no CPU fallback")
        sys.exit(2)
    L.fasp_hip_set_device(local_rank)
    # Device work is issued by libfasp_hip on its own HIP stream; torch is not involved in
    # the single-GPU path, so the synchronisation bracket is the library's own stream sync
    # (torch.cuda.synchronize() would only see torch's idle default stream).

    n = args.n
    t0 = time.perf_counter()
    ia, ja, a, f, ue = fa.poisson7pt(n)
    m, nnz = len(f), len(a)
    log(f"P7({n}): {m} rows, {nnz} nnz, generated in {time.perf_counter()-t0:.2f} s")

    itp, amgp = workload_params()

    def sync():
        L.fasp_hip_device_synchronize()

    def timed_solves(Hx, warmup, steps):
        st = hist = stats = None
        for _ in range(warmup):
            st, hist, stats = Hx.solve_resident(itp)
        sync()
        t0 = time.perf_counter()
        sp = []
        for _ in range(steps):
            st, hist, stats = Hx.solve_resident(itp)
            sp.append(stats.spmv_ms)
        sync()
        return time.perf_counter() - t0, st, hist, stats, float(np.mean(sp))

    if args.only_variable:   # the profiled command of profiles/r04_rocprof_var (tools/profile.sh <tag> variable)
        print(json.dumps({"variable_coefficient": variable_leg(n, (ia, ja, a, f, ue), itp, amgp, timed_solves)}), flush=True)
        return

    t0 = time.perf_counter()
    H = fa.AMG(ia, ja, a, amgp)
    t_setup = time.perf_counter() - t0
    H.set_rhs(f)
    log(f"AMG setup + upload: {t_setup:.2f} s, {H.num_levels} levels")

    elapsed, st, hist, stats, kernel_ms = timed_solves(H, args.warmup, args.steps)
    ms_per_step = 1e3 * elapsed / args.steps
    value = m * args.steps / elapsed
    x = H.get_solution()
    log(f"solve: {st} iterations, relres {stats.relres:.10e}, {ms_per_step:.2f} ms/solve, "
        f"coarse its {stats.coarse_iters}, max|x-u_exact| {np.max(np.abs(x-ue)):.3e}")

    B = spmv_bytes(m, m, nnz)
    kind, matrix_bytes = H.kernel_info(0, 0)
    moved = matrix_bytes + 8.0 * m + 8.0 * m   # stored matrix form + x once + y once (the dotted vector is x)
    coded = kind in (4, 5, 6)
    roof = roofline_entry(kind, moved, kernel_ms, int(stats.spmv_launches) * args.steps, workload="constant", n=n,
                          note=("the operator is stored losslessly coded (2 bytes per row + a 27-entry pattern table): "
                                "bytes_per_launch is what this kernel has to move, not SURVEY 8(d)'s plain-CSR figure "
                                f"({B} B); the plain-CSR kernel of the same operator is timed in the same run: "
                                "roofline_plain_csr") if coded else None)
    roof["plain_csr_algorithmic_bytes"] = B

    # The same operator through the plain-CSR kernel, timed the same way as the coded one: INSIDE a solve (coding switched off for
    # one warm-up and two timed solves: every coded operator of the hierarchy then runs its plain-CSR kernel; HIP events around the
    # level-0 t = A p launches).  SURVEY 8(d)'s bytes over that time = north_star's "SpMV >= 60 % of the HBM roofline".
    plain = None
    if coded and not args.no_plain:
        try:
            L.fasp_hip_tune(b"compress", 0)
            pkind, _pb = H.kernel_info(0, 0)
            el_p, st_p, _hp, stats_p, ms_plain = timed_solves(H, 1, 2)
            L.fasp_hip_tune(b"compress", 1)
            plain = roofline_entry(pkind, float(B), ms_plain, int(stats_p.spmv_launches) * 2, workload="constant", n=n)
            # the profiled command of this workload runs the coded operators only (tools/profile.sh: --no-plain keeps the solve-wide sum
            # clean); the same kernel instantiation on the same grid and sparsity pattern is in this round's variable-coefficient pass
            cur = TRAFFIC_JSONS["constant"][0]
            if plain.get("traffic_source") != cur:
                name_p = KERNELS.get(pkind, (str(pkind),))[0]
                tr, src = pmc_traffic(name_p, "variable", n)
                if tr is not None and src == TRAFFIC_JSONS["variable"][0]:
                    plain["traffic"] = tr
                    plain["traffic_source"] = src + " (same kernel instantiation, grid and sparsity pattern; the variable-coefficient values)"
                    plain["traffic_GBps"] = tr / (ms_plain * 1e-3) / 1e9 if ms_plain > 0 else None
                    plain["traffic_over_bytes"] = tr / float(B)
            plain["timed"] = "inside two solves with every coded operator on its plain-CSR kernel (HIP events around the level-0 launches)"
            plain["solve_ms_per_step_all_plain_csr"] = 1e3 * el_p / 2
            plain["iterations"] = int(st_p)
            log(f"plain-CSR level-0 SpMV in-solve: {ms_plain*1e3:.1f} us = {plain['achieved']:.0f} GB/s = {plain['frac']:.3f} of peak; "
                f"solve with plain CSR everywhere {1e3*el_p/2:.2f} ms, {st_p} iterations")
        except Exception as e:
            L.fasp_hip_tune(b"compress", 1)
            log(f"plain-CSR timing failed: {e!r}")
    elif not coded:
        plain = dict(roof)
    # inside the object the driver keeps: what SURVEY 8(d) asks for next to what the coded kernel does
    roof["frac_by_survey_8d_bytes"] = (B / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if kernel_ms > 0 else None
    if coded:
        roof["frac_by_survey_8d_bytes_note"] = ("not comparable (> 1): the coded operator does not read IA / JA / val; the plain-CSR kernel of "
                                                "the same operator, same timing method: plain_csr")
    roof["plain_csr"] = ({k: plain[k] for k in ("kernel", "achieved", "frac", "bytes_per_launch", "ms_per_launch", "launches_timed", "traffic",
                                                "traffic_source", "traffic_over_bytes", "timed", "solve_ms_per_step_all_plain_csr") if k in plain}
                         if plain else None)
    # solve-wide: HBM-side bytes per PCG iteration (PMC passes of this command, profiles/<round>_rocprof/traffic.json) over the time of
    # an iteration in THIS run
    sw = solve_wide_traffic("constant", n)
    if sw and st > 0:
        it_ms = ms_per_step / int(st)
        roof["solve_wide"] = {"bytes_per_iteration": sw[0], "ms_per_iteration": it_ms, "achieved": sw[0] / (it_ms * 1e-3) / 1e9,
                              "frac": sw[0] / (it_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "unit": "GB/s", "traffic_source": sw[1]}
    else:
        roof["solve_wide"] = None

    ceilings = None
    try:
        out3 = (C.c_double * 3)()
        if not args.no_ceilings and L.fasp_hip_measure_ceilings(out3, C.c_size_t(1 << 30), 5) == 0:
            ceilings = {"unit": "GB/s", "read": out3[0], "copy": out3[1], "triad": out3[2],
                        "buffer_bytes": 1 << 30,
                        "note": "16 bytes per lane, 1024-block grid, HIP events; roofline fractions use the nominal 8000 GB/s",
                        "device_state": device_state()}
            log(f"device ceilings: read {out3[0]:.0f}, copy {out3[1]:.0f}, triad {out3[2]:.0f} GB/s; {ceilings['device_state']}")
    except Exception as e:
        log(f"ceiling measurement failed: {e!r}")

    out = {
        "metric": f"AMG-PCG solve DOF/s (3D 7-pt Poisson {n}^3, classical AMG V(1,1) w-Jacobi + PCG, rtol 1e-8)",
        "value": value, "unit": "DOF/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"P7({n}): 3-D 7-point Poisson {n}^3, {m} DOF, {nnz} nnz; "
                               "PCG rtol 1e-8 + classical RS-AMG V(1,1), Jacobi w=0.6667; "
                               "one step = one full solve on the resident hierarchy",
                   "rows": m, "nnz": nnz, "levels": H.num_levels, "parallelism": "1 GPU"},
        "iterations": int(st), "relres": stats.relres,
        "setup_seconds": t_setup,
        "roofline": roof,
        "roofline_plain_csr": plain,
        "ceilings": ceilings,
    }
    # The parity statement of the line: against the REFERENCE's own run of this solve (tests/golden/p7_scale.npz, written by
    # tools/gen_golden_f5*.py from the compiled reference; a committed fixture, not the oracle): iteration count, final
    # relative residual (absolute bar 1e-10; relative deviation printed beside it), the residual history.
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "p7_scale.npz"))
        if f"n{n}_iters" in z.files:
            it_ref, rr_ref = int(z[f"n{n}_iters"]), float(z[f"n{n}_relres"])
            pr = {"iters_gpu": int(st), "iters_reference": it_ref, "relres_gpu": stats.relres, "relres_reference": rr_ref,
                  "abs_dev_relres": abs(stats.relres - rr_ref), "rel_dev_relres": abs(stats.relres - rr_ref) / rr_ref,
                  "source": "compiled reference (tests/golden/p7_scale.npz)"}
            ok = int(st) == it_ref and abs(stats.relres - rr_ref) <= 1e-10 and abs(stats.relres - rr_ref) <= 1e-6 * rr_ref
            if f"n{n}_hist" in z.files:
                hr = z[f"n{n}_hist"]
                hd = np.concatenate([hist[:-2], hist[-1:]])   # (the device history ends with recurrence and true residual of the last iteration)
                if len(hd) == len(hr):
                    pr["max_rel_dev_residual_history"] = float(np.max(np.abs(hd[:-1] - hr[:-1]) / hr[:-1]))
                    ok = ok and pr["max_rel_dev_residual_history"] <= 1e-8
                else:
                    ok = False
            pr["ok"] = bool(ok)
            out["parity_reference"] = pr
    except Exception as e:
        log(f"parity_reference unavailable: {e!r}")
    if not args.no_cpu_baseline:
        try:
            cb, its_cpu, rr_cpu, hist_dev = cpu_baseline(H, ia, ja, a, f, int(st), hist,
                                                         float(os.environ.get("BENCH_CPU_BUDGET_S", "15")), baseline_candidates())
            out["cpu_baseline"] = cb
            out["parity"] = {"against": "the multi-thread CPU baseline of this run (the oracle in its TIMING mode: OpenMP-regrouped "
                                        "reductions -- not the reference's serial sums; the reference itself: parity_reference)",
                             "iters_gpu": int(st), "iters_cpu": its_cpu, "relres_gpu": stats.relres,
                             "relres_cpu": rr_cpu, "max_rel_dev_residual_history": hist_dev}
            cb1, _i, _r, _h = cpu_baseline(H, ia, ja, a, f, int(st), hist, 0.0, 1)
            out["cpu_baseline_1thread"] = cb1
        except Exception as e:  # the baseline is a report, never a reason to lose the line
            log(f"cpu_baseline failed: {e!r}")
            out.setdefault("cpu_baseline", None)
    H.close()

    if not args.no_variable:
        out["variable_coefficient"] = variable_leg(n, (ia, ja, a, f, ue), itp, amgp, timed_solves)
        if out["variable_coefficient"]:   # the general-matrix number inside the keys the driver keeps
            vc = out["variable_coefficient"]
            out["roofline"]["variable_coefficient"] = {"ms_per_step": vc["ms_per_step"], "iterations": vc["iterations"],
                                                       "level0_spmv_frac": vc["roofline"]["frac"], "level0_kernel": vc["roofline"]["kernel"]}
            out["config"]["variable_coefficient_ms_per_step"] = vc["ms_per_step"]
    if not args.no_extra and args.gpus == 1:
        del ia, ja, a
        out["other_configs"] = other_configs_leg()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
