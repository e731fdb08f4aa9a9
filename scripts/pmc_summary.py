"""Summarise rocprofv3 --pmc counter_collection CSVs: median counter value per kernel (full-batch dispatches)."""
import collections
import csv
import glob
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(d + "/pass*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "tq_" not in name:
            continue
        name = name.split("(")[0].replace("void ", "")
        agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), ctrs in sorted(agg.items()):
    print(f"{name}  grid={grid}")
    for c, v in sorted(ctrs.items()):
        v = sorted(v)
        print(f"    {c:28s} {v[len(v)//2]:16.0f}   (n={len(v)})")
