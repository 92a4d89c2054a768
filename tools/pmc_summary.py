#!/usr/bin/env python3
"""Mean per-launch value of every counter in a rocprofv3 --pmc output directory, for kernels matching a substring."""
import collections, csv, glob, sys
d, kern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_relax2")
for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in sorted(acc.items()):
        print("  %-36s %.5g   (n=%d)" % (k, sum(v) / len(v), len(v)))
