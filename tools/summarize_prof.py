"""Summarise rocprofv3 output of tools/profile.sh: per-kernel count / mean duration from the
kernel trace, and per-kernel FETCH_SIZE / WRITE_SIZE per launch from the two PMC passes."""
import csv, glob, os, sys, collections

root = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "constant"
command = sys.argv[3] if len(sys.argv) > 3 else "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variable"
N = int(os.environ.get("BENCH_N", "256"))

def find(sub, pat):
    g = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return g[0] if g else None

def short(name):
    name = name.replace("fasp::", "")
    if "(" in name: name = name[:name.index("(")]
    return name.replace("void ", "")[:70]

trace = find("trace", "*kernel_trace.csv")
dur = collections.defaultdict(list)
if trace:
    for row in csv.DictReader(open(trace)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
def pmc(sub, counter):
    f = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc
fetch = pmc("pmc_fetch", "FETCH_SIZE")
write = pmc("pmc_write", "WRITE_SIZE")
tot = sum(sum(v) for v in dur.values())
print(f"# rocprofv3 summary (python3 {command}; workload: {workload} coefficients, n = {N})\n")
print("| kernel | launches | mean us | total ms | % | FETCH_SIZE KB/launch (raw) | WRITE_SIZE KB/launch |")
print("|---|---|---|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    fs = fetch.get(k); ws = write.get(k)
    print(f"| {k} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.1f} | "
          f"{(sum(fs)/len(fs)) if fs else float('nan'):.0f} | {(sum(ws)/len(ws)) if ws else float('nan'):.0f} |")
# HBM-side traffic per launch of every OP_MXV_DOT (= 7) row kernel and of the streaming calibration kernels:
# FETCH_SIZE is doubled (gfx950 reports 1/2 of the bytes of wide coalesced reads, MI355X_MICROARCH.md; the
# k_dot / k_norms rows below, whose byte counts are known, show the factor), WRITE_SIZE is taken as is; both are
# KB per launch from SEPARATE --pmc passes.  Only the level-0 launches (the largest half) are averaged.
import json, re
kernels = {}
for k, v in dur.items():
    if not ("k_csr" in k and re.search(r"<7,|<7>|, 7>", k)) and k not in ("k_dot", "k_norms", "k_cg_update", "k_axpby"):
        continue
    fs, ws = fetch.get(k), write.get(k)
    if not fs or not ws:
        continue
    top = sorted(fs, reverse=True)[:max(1, len(fs) // 2)]
    topw = sorted(ws, reverse=True)[:max(1, len(ws) // 2)]
    topd = sorted(v, reverse=True)[:max(1, len(v) // 2)]
    kernels[k] = {"bytes_per_launch": (2.0 * sum(top) / len(top) + sum(topw) / len(topw)) * 1024.0,
                  "fetch_KB_raw": sum(top) / len(top), "write_KB": sum(topw) / len(topw),
                  "mean_us_largest_half": sum(topd) / len(topd), "launches": len(v)}
# solve-wide: every kernel of the solves (setup / upload kernels, fills and copies of the runtime, and the ceiling kernels left out),
# 2 x FETCH_SIZE + WRITE_SIZE summed, over the PCG iterations of the run (k_cg_update runs once per iteration)
def solve_kernel(k):
    return not ("rocclr" in k or "k_sort_rows" in k or "k_read16" in k or "k_copy16" in k or "k_triad16" in k or k.startswith("k_reorder") or k.startswith("k_permute"))
iters = len(dur.get("k_cg_update", []))
fsum = sum(sum(v) for k, v in fetch.items() if solve_kernel(k))
wsum = sum(sum(v) for k, v in write.items() if solve_kernel(k))
tsum = sum(sum(v) for k, v in dur.items() if solve_kernel(k))
solve_wide = None
if iters and fsum and wsum:
    solve_wide = {"iterations": iters, "bytes_per_iteration": (2.0 * fsum + wsum) * 1024.0 / iters, "kernel_us_per_iteration": tsum / iters,
                  "note": "sum over all solve kernels of 2 x FETCH_SIZE + WRITE_SIZE (separate passes) / PCG iterations (= k_cg_update launches)"}
json.dump({"command": "python3 " + command, "workload": workload, "n": N, "solve_wide": solve_wide,
           "note": "HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes); level-0 launches",
           "kernels": kernels}, open(os.path.join(root, "traffic.json"), "w"), indent=1)
if solve_wide:
    print(f"\n## solve-wide\n\n- {solve_wide['iterations']} PCG iterations, {solve_wide['bytes_per_iteration']/1e9:.2f} GB of HBM-side traffic and "
          f"{solve_wide['kernel_us_per_iteration']:.0f} us of kernel time per iteration -> {solve_wide['bytes_per_iteration']/solve_wide['kernel_us_per_iteration']/1e3:.0f} GB/s "
          f"= {solve_wide['bytes_per_iteration']/solve_wide['kernel_us_per_iteration']/1e3/8000:.2f} of 8 TB/s (profiled clocks)")
print("\n## traffic per launch (level-0 launches)\n")
for k, r in kernels.items():
    print(f"- {k}: {r['bytes_per_launch']/1e9:.3f} GB ({r['fetch_KB_raw']:.0f} KB FETCH_SIZE raw x 2 + {r['write_KB']:.0f} KB WRITE_SIZE), "
          f"{r['mean_us_largest_half']:.1f} us -> {r['bytes_per_launch']/r['mean_us_largest_half']/1e3:.0f} GB/s")
