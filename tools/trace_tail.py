#!/usr/bin/env python3
"""Print the last N dispatches of a rocprofv3 kernel trace (name, duration, gap to the previous one) and write them as csv.
usage: trace_tail.py <p_kernel_trace.csv> [N = 200] [out.csv]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(tail)
prev = None
agg = {}
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-48:]
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += ((s - prev) / 1e3 if prev else 0.0)
    prev = e
for k, (c, d, g) in agg.items():
    print(f"{k:50s} n {c:4d}  avg dur {d / c:6.2f} us  avg gap before {g / c:6.2f} us")
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
print(f"{len(tail)} dispatches in {(t1 - t0) / 1e3:.1f} us")
