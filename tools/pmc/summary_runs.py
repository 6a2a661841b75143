"""Counter means per RUN of consecutive dispatches of one kernel (runs are separated by k_dot launches):
python3 tools/pmc/summary_runs.py <rocprofv3 output dir> ...   (development tool)"""
import collections
import csv
import glob
import os
import sys

table = collections.OrderedDict()
for root in sys.argv[1:]:
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.defaultdict(dict)   # dispatch -> {counter: value}, name
    name = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        k = r["Kernel_Name"].replace("fasp::", "").replace("void ", "")
        name[d] = k[:k.index("(")] if "(" in k else k
    run, prev = -1, None
    for d in sorted(per):
        k = name[d]
        if k != prev:
            run += 1
            prev = k
        if "k_dot" in k or "k_finalize" in k or "rocclr" in k:
            continue
        key = (run, k)
        e = table.setdefault(key, collections.defaultdict(list))
        for c, v in per[d].items():
            e[c].append(v)
for (run, k), e in table.items():
    print(f"run {run:3d} {k}")
    for c in sorted(e):
        v = e[c]
        print(f"      {c:28s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
