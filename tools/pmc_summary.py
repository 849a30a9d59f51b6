"""Summarise rocprofv3 --pmc counter_collection.csv: mean counter values per kernel name (selected kernels)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    keys = sys.argv[2:] or ("conv2d_bf16x3_kernel<3>", "spconv_fwd_rows<64, 64>", "spconv_fwd_rows<32, 32>")
    if not any(k in name for k in keys):
        continue
    agg[name[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in agg.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
