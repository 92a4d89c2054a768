#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another; see MI355X_MICROARCH.md
"HBM" / "rocprofv3 PMC slots") over tools/prof_sweep.py into profiles/traffic.json: HBM bytes per K2 launch.

gfx950 corrections applied as that guide prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.

A third, optional pass with the SQ counters (SQ_INSTS_VALU ...) adds the VALU wave-instructions per launch: the pass the bench
times is bound by fp64 issue, and bench.py prices its launch time against the issue peak as well (roofline.valu_issue).

usage: pmc_traffic.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run> <n> <out.json> [kernel substring = k_relax3] [sweeps per launch = 3] [dir with SQ run]"""
import csv
import datetime
import glob
import hashlib
import json
import os
import sys


def mean_counter(d, name, kernel="k_apply"):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)


def mean_duration_us(d, kernel="k_apply"):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    return sum(ds) / len(ds)


if __name__ == "__main__":
    dfetch, dwrite, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    kern = sys.argv[5] if len(sys.argv) > 5 else "k_relax3"
    spl = int(sys.argv[6]) if len(sys.argv) > 6 else {"k_relax2": 2, "k_relax3": 3}.get(kern, 1)
    fetch_kib, nf = mean_counter(dfetch, "FETCH_SIZE", kern)
    write_kib, nw = mean_counter(dwrite, "WRITE_SIZE", kern)
    read_bytes = 2.0 * fetch_kib * 1024.0
    write_bytes = write_kib * 1024.0
    res = {
        "n": n, "kernel": kern, "sweeps_per_launch": spl, "launches_sampled": [nf, nw],
        "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
        "hbm_read_bytes_per_launch": read_bytes, "hbm_write_bytes_per_launch": write_bytes,
        "hbm_bytes_per_launch": read_bytes + write_bytes, "algorithmic_bytes_per_launch": 32.0 * n * n * spl,
        "ratio_to_algorithmic": (read_bytes + write_bytes) / (32.0 * n * n * spl),
        "avg_kernel_us_under_pmc": [mean_duration_us(dfetch, kern), mean_duration_us(dwrite, kern)],
        "kernels_hash": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "turbomesh_amd", "csrc",
                                                         "tm_kernels.hip"), "rb").read()).hexdigest()[:16],
        "date": datetime.date.today().isoformat(),
        "corrections": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read), KiB -> bytes; separate --pmc passes",
    }
    if len(sys.argv) > 7 and os.path.isdir(sys.argv[7]):
        try:
            insts, ns = mean_counter(sys.argv[7], "SQ_INSTS_VALU", kern)
            res["valu_wave_insts_per_launch"] = insts
            res["valu_launches_sampled"] = ns
            res["avg_kernel_us_under_pmc"].append(mean_duration_us(sys.argv[7], kern))
        except (IndexError, ZeroDivisionError):
            pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
