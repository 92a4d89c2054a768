#!/usr/bin/env python3
"""Per-kernel launch statistics of a rocprofv3 kernel trace WITHOUT the clock ramp of a cold device.

`rocprofv3 --stats` averages every launch of a run.  A bench run starts on a cold device and the power controller needs ~50 ms of load
to settle the shader clock (tools/dev/ramp_probe.py), so the first few hundred launches of the sweep kernel are 5-30 % slower than
the timed region and pull the `--stats` average up by 2-4 %.  This tool reads the kernel TRACE of the same run and, per kernel,
drops the launches that START within `skip_ms` of that kernel's first launch; mean / median / p10 / p90 of the rest are what
`roofline.achieved` of the bench line must reproduce (bytes per launch / mean_us_steady).

usage: steady_stats.py <p_kernel_trace.csv> [out.csv] [--skip-ms 60] [--min-calls 8] [--tail N]
  --tail N   additionally report the mean of the LAST N launches of every kernel (N = launches of the timed region)"""
import csv
import statistics
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    opts = {"--skip-ms": 60.0, "--min-calls": 8, "--tail": 0}
    pos, it = [], iter(sys.argv[1:])
    for a in it:
        if a in opts:
            opts[a] = float(next(it))
        else:
            pos.append(a)
    if not pos:
        raise SystemExit(__doc__)
    skip_ns = opts["--skip-ms"] * 1e6
    tail = int(opts["--tail"])
    per = {}
    for r in csv.DictReader(open(pos[0])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        per.setdefault(r["Kernel_Name"], []).append((s, e - s))
    rows = []
    for name, launches in per.items():
        launches.sort()
        if len(launches) < opts["--min-calls"]:
            continue
        t_first = launches[0][0]
        all_d = [d for _, d in launches]
        steady = [d for s, d in launches if s - t_first >= skip_ns] or all_d
        q = statistics.quantiles(steady, n=10) if len(steady) >= 10 else [min(steady)] * 9
        row = {"Name": name, "Calls": len(all_d), "AverageNs_all": sum(all_d) / len(all_d), "CallsSteady": len(steady),
               "AverageNs_steady": sum(steady) / len(steady), "MedianNs_steady": statistics.median(steady), "P10Ns_steady": q[0], "P90Ns_steady": q[8],
               "MinNs": min(all_d), "MaxNs": max(all_d), "SkipMs": opts["--skip-ms"], "TotalNs_all": sum(all_d)}
        if tail:
            t = all_d[-tail:]
            row["TailCalls"] = len(t)
            row["AverageNs_tail"] = sum(t) / len(t)
        rows.append(row)
    rows.sort(key=lambda r: -r["TotalNs_all"])
    if len(pos) > 1:
        with open(pos[1], "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
    for r in rows[:12]:
        extra = f"  last {r['TailCalls']}: {r['AverageNs_tail'] / 1e3:8.2f}" if tail else ""
        print(f"{short(r['Name'])[:60]:60s} calls {r['Calls']:6d}  avg all {r['AverageNs_all'] / 1e3:8.2f} us | steady ({r['CallsSteady']} calls) mean {r['AverageNs_steady'] / 1e3:8.2f} "
              f"median {r['MedianNs_steady'] / 1e3:8.2f} p10 {r['P10Ns_steady'] / 1e3:8.2f} p90 {r['P90Ns_steady'] / 1e3:8.2f}{extra}")


if __name__ == "__main__":
    main()
