"""Per kernel: median duration, GRBM_GUI_ACTIVE / 8 (cycles) and the clock they imply, over the last half of the dispatches."""
import csv, glob, sys, statistics, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "fa2" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            rows[name].append((float(r["Counter_Value"]) / 8, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, v in rows.items():
    v = v[len(v) // 2:]
    cyc, us = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
    print(f"{k:42s} n={len(v):3d}  {us:9.1f} us  {cyc/1e6:8.3f} Mcycles  {cyc/us/1e3:6.3f} GHz")
