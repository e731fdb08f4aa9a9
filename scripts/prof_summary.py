"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and grid size, median/min/max duration."""
import collections
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/bench_kernel_trace.csv"
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if "tq_" not in name:
        continue
    name = name.split("(")[0].replace("void ", "")
    agg[(name, int(r["Grid_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"])].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':52s} {'grid':>9s} {'vgpr':>5s} {'lds':>6s} {'n':>4s} {'med_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for k, v in sorted(agg.items()):
    v = sorted(v)
    print(f"{k[0]:52s} {k[1]:9d} {k[2]:>5s} {k[3]:>6s} {len(v):4d} {v[len(v)//2]:9.1f} {v[0]:9.1f} {v[-1]:9.1f}")
