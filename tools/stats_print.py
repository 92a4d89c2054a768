#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 --stats kernel_stats.csv: calls, average / total microseconds, share.
usage: stats_print.py <kernel_stats.csv> [rows = 15]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:9.1f} total_us {float(r['TotalDurationNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}%")
