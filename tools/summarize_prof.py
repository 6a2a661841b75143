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
# the largest launches of the level-0 SpMV kernel and the calibration kernel
def top(acc, key, n=3):
    for k, v in acc.items():
        if key in k:
            s = sorted(v, reverse=True)[:n]
            return k, s
    return None, []
for key in ("OP_MXV_DOT", "k_csr_wstream<7", "k_norms", "k_dot"):
    k, s = top(fetch, key)
    k2, s2 = top(write, key)
    kd, sd = top(dur, key)
    print(f"\n{key}: fetch top {s} KB ({k}); write top {s2} KB; longest launches {sd} us")
