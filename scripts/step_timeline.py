"""Kernel timeline (start offset, duration, gap to the previous kernel) of a few consecutive full-batch steps in a
rocprofv3 --kernel-trace CSV, all kernels included (RCCL's too)."""
import csv
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tq_ksmogn_il2_kernel" in r["Kernel_Name"] and "true" in r["Kernel_Name"]]
mid = idx[len(idx) // 2]
lo = mid - 9
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
for r in rows[lo:mid + 16]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:50]:50s} q={r['Queue_Id']:>3s} dur={(e - s) / 1e3:7.1f} gap={gap:7.1f}")
    prev_end = max(prev_end or 0, e)
