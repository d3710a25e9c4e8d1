"""Summarises rocprofv3 --pmc counter_collection CSVs: per kernel, mean of each counter."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fa2" not in k: continue
        name = k.split("(")[0].split("::")[-1][:28]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
