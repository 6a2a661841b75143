"""Per-kernel mean of every counter in a rocprofv3 --pmc ... --output-format csv directory (dev tool)."""
import csv, glob, os, sys, collections
for root in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("fasp::", "").replace("void ", "")
            k = k[:k.index("(")] if "(" in k else k
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if "k_csr" in k or "k_copy" in k or "k_read" in k or "k_triad" in k or "spcg" in k:
            print(f"{os.path.basename(root):28s} {k:40s} {c:12s} n={len(v):3d} mean={sum(v)/len(v):14.1f} max={max(v):14.1f}")
