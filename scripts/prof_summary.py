"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and grid size, median/min/max duration.
usage: prof_summary.py trace.csv [--last N]     (--last N: only the last N launches of each kernel, e.g. the trained regime
at the end of a bench run)"""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0] if args else "gpurun_out/prof/bench_kernel_trace.csv"
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if "tq_" not in name:
        continue
    name = name.split("(")[0].replace("void ", "")
    agg[(name, int(r["Grid_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"])].append(
        (int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(f"{'kernel':52s} {'grid':>9s} {'vgpr':>5s} {'lds':>6s} {'n':>5s} {'med_us':>9s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for k, v in sorted(agg.items()):
    v = [d for _, d in sorted(v)]
    if last:
        v = v[-last:]
    v = sorted(v)
    print(f"{k[0]:52s} {k[1]:9d} {k[2]:>5s} {k[3]:>6s} {len(v):5d} {v[len(v)//2]:9.1f} {sum(v)/len(v):9.1f} {v[0]:9.1f} {v[-1]:9.1f}")
