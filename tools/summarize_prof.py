"""Summarise rocprofv3 output of tools/profile.sh: per-kernel count / mean duration from the
kernel trace, and per-kernel FETCH_SIZE / WRITE_SIZE per launch from the two PMC passes."""
import csv, glob, os, sys, collections

root = sys.argv[1]

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
print("# rocprofv3 summary (bench.py --steps 3 --warmup 1 --no-cpu-baseline, P7(256))\n")
print("| kernel | launches | mean us | total ms | % | FETCH_SIZE KB/launch (raw) | WRITE_SIZE KB/launch |")
print("|---|---|---|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    fs = fetch.get(k); ws = write.get(k)
    print(f"| {k} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.1f} | "
          f"{(sum(fs)/len(fs)) if fs else float('nan'):.0f} | {(sum(ws)/len(ws)) if ws else float('nan'):.0f} |")
# traffic of the roofline kernel (level-0 t = A p fused with (t,p)): the OP_MXV_DOT (= 7) instantiation
# with the largest mean duration.  FETCH_SIZE is doubled (gfx950 reports 1/2 of the bytes of wide
# coalesced reads, MI355X_MICROARCH.md; calibrated here on k_dot / k_norms whose byte counts are
# known), WRITE_SIZE is taken as is; both are KB per launch.
import json, re
cands = [(sum(v) / len(v), k) for k, v in dur.items() if re.search(r"<7,|<7>|, 7>", k) and "k_csr" in k]
if cands:
    _, kname = max(cands)
    fs, ws = fetch.get(kname), write.get(kname)
    if fs and ws:
        top = sorted(fs, reverse=True)[:max(1, len(fs) // 2)]   # level-0 launches are the largest
        topw = sorted(ws, reverse=True)[:max(1, len(ws) // 2)]
        bytes_per_launch = (2.0 * sum(top) / len(top) + sum(topw) / len(topw)) * 1024.0
        kind = 5 if "rowpat" in kname else 4 if "dict8" in kname else 2 if "wstream" in kname else 0
        json.dump({"kernel": kname, "kernel_kind": kind, "bytes_per_launch": bytes_per_launch,
                   "fetch_KB_raw": sum(top) / len(top), "write_KB": sum(topw) / len(topw),
                   "mean_us": sum(dur[kname]) / len(dur[kname]),
                   "note": "HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (separate --pmc passes)"},
                  open(os.path.join(root, "traffic.json"), "w"), indent=1)

# the largest launches of the level-0 SpMV kernel and the calibration kernel
def top(acc, key, n=3):
    for k, v in acc.items():
        if key in k:
            s = sorted(v, reverse=True)[:n]
            return k, s
    return None, []
for key in ("k_csr_rowpat<7", "k_csr_wstream<7", "k_norms", "k_dot"):
    k, s = top(fetch, key)
    k2, s2 = top(write, key)
    kd, sd = top(dur, key)
    print(f"\n{key}: fetch top {s} KB ({k}); write top {s2} KB; longest launches {sd} us")
