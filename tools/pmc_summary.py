"""Summarises rocprofv3 --pmc counter_collection CSVs: per kernel, mean of each counter.

    python tools/pmc_summary.py <dir with the passes' output> [--json profiles/pmc_traffic.json --source <name>]

With --json it also writes the per-launch HBM traffic table bench.py reads for `roofline.traffic`:
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md prescribes
(HBM section), both counters are in units of 1024 B as rocprofv3 prints them."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fa2" not in k:
            continue
        name = k.split("(")[0].split("::")[-1][:28]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    src = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else root
    tab = {"source": src, "unit": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 at (B=4,H=16,N=8192,d=128)", "kernels": {}}
    for k, cs in acc.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            base = k.split("<")[0]
            f, w = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            tab["kernels"][base] = {"bytes_per_launch": int((2 * f + w) * 1024), "FETCH_SIZE": f, "WRITE_SIZE": w}
    json.dump(tab, open(out, "w"), indent=1)
    print("wrote", out)
