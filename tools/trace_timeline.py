#!/usr/bin/env python3
"""Prints the tail of a rocprofv3 kernel trace (csv) as a timeline: start (us, relative), duration, gap to the previous END on any
queue, queue id, kernel name.  usage: trace_timeline.py <kernel_trace.csv> [last N = 40]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name}")
    prev_end = max(prev_end, e)
