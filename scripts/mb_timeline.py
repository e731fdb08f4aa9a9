"""Timeline of the minibatch steps in a rocprofv3 --kernel-trace CSV: per launch start offset, duration, gap to the
previous launch (small-grid launches only)."""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/bench_kernel_trace.csv"
rows = [r for r in csv.DictReader(open(path)) if "tq_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the minibatch phase: launches of the 16-lane pixel kernel (tq_ksmogn_kernel) mark it
idx = [i for i, r in enumerate(rows) if "tq_ksmogn_kernel" in r["Kernel_Name"]]
if not idx:
    sys.exit("no minibatch launches found")
lo, hi = idx[len(idx) // 2] - 12, idx[len(idx) // 2] + 14
prev_end = None
for r in rows[max(lo, 0):hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]:44s} grid={int(r['Grid_Size_X']):8d} dur={(e - s) / 1e3:7.1f} us gap={gap:7.1f} us")
    prev_end = e
