"""Mean of every collected PMC counter per kernel name over the rocprofv3 counter_collection.csv files below a directory.
usage: python tools/pmc_kernels.py <dir> [kernel-substring ...]"""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if subs and not any(s in name for s in subs):
            continue
        name = re.sub(r"\(.*", "", name)[:70]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
