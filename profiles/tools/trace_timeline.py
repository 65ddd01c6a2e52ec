"""Print the kernel timeline of a few consecutive steps from a rocprofv3 --kernel-trace CSV: name, start offset,
duration, idle gap before it (all us).  usage: trace_timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(e, prev_end or 0)
