"""Per (segment, kernel) means of every counter collected by tools/pmc/run_bound.sh, one row per segment with the counters
of all passes side by side:  python3 tools/pmc/bound_summary.py <dir with g*/ passes and g*.log>   (development tool)"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
FLUSH_GRID = 1024 * 256
seginfo = {}
for log in sorted(glob.glob(os.path.join(root, "g*.log"))):
    for line in open(log):
        m = re.match(r"segment (\d+) (level \d+ op \d+ kernel-kind \d+ bytes \d+): ([\d.]+) us", line)
        if m:
            seginfo.setdefault(int(m.group(1)), (m.group(2), []))[1].append(float(m.group(3)))
table = collections.OrderedDict()   # (seg, kernel) -> counter -> [values]
for p in sorted(glob.glob(os.path.join(root, "g*"))):
    if not os.path.isdir(p):
        continue
    per = collections.defaultdict(dict); name = {}; grid = {}; dur = {}
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            k = r["Kernel_Name"].replace("fasp::", "").replace("void ", "")
            name[d] = k[:k.index("(")] if "(" in k else k
            grid[d] = int(r["Grid_Size"])
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seg, in_sep, started = 0, False, False
    for d in sorted(per):
        k = name[d]
        if "k_dot" in k:
            if started and not in_sep:
                seg += 1
            in_sep = True
            continue
        if "rocclr" in k or "k_finalize" in k or "k_sort" in k:
            continue
        if k.startswith("k_read16") and grid[d] == FLUSH_GRID:
            started = True   # the first flush opens segment 0
            in_sep = False
            continue
        if not started:
            continue
        in_sep = False
        e = table.setdefault((seg, k), collections.defaultdict(list))
        for c, v in per[d].items():
            e[c].append(v)
        e["_us_profiled"].append(dur[d])
for (seg, k), e in table.items():
    info = seginfo.get(seg, ("?", [0.0]))
    us = sum(info[1]) / max(1, len(info[1]))
    print(f"segment {seg:3d} {info[0]}  {k}   event-timed {us:.1f} us")
    for c in sorted(e):
        v = e[c][1:] if len(e[c]) > 1 else e[c]    # the first launch of a segment is the untimed warm-up
        print(f"      {c:40s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
