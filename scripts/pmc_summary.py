"""Per-kernel averages of the counters of one rocprofv3 --pmc pass.  usage: pmc_summary.py <dir> [name-filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if flt in k:
        agg[(k[:60], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in agg.items():
    print(k, g)
    for c, v in cs.items():
        print(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
