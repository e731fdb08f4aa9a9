"""profiles/rNN_kernel_trace.json: per config, average / median duration of every tq_ kernel in a rocprofv3 --kernel-trace CSV
of `bench.py` (full-size grids only: the largest grid of each kernel).
usage: make_trace_json.py out.json cfg=trace.csv[:last N] ..."""
import collections
import csv
import json
import sys

out = sys.argv[1]
res = {}
for arg in sys.argv[2:]:
    cfg, rest = arg.split("=", 1)
    path, _, last = rest.partition(":")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "tq_" not in name:
            continue
        name = name.split("(")[0].replace("void ", "")
        agg[name].append((int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    res[cfg] = {}
    for name, v in agg.items():
        gmax = max(g for _, g, _ in v)
        d = [x for _, g, x in sorted(v) if g == gmax]
        if last:
            d = d[-int(last):]
        ds = sorted(d)
        res[cfg][name] = {"grid": gmax, "n": len(d), "avg_us": sum(d) / len(d), "med_us": ds[len(ds) // 2], "min_us": ds[0], "max_us": ds[-1]}
json.dump(res, open(out, "w"), indent=1)
