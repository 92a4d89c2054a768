#!/bin/bash
# HBM traffic and VALU share of the two kernels of the BiCGStab iteration at 4096^2 (separate --pmc passes, kernel trace only):
#   usage (through gpurun, from the repo root): tools/dev/pmc_krylov.sh  -> gpurun_out/pmc_krylov/summary.txt
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_krylov
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o p -- python3 "$root/tools/prof_sweep.py" 4096 12 bicgstab > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o p -- python3 "$root/tools/prof_sweep.py" 4096 12 bicgstab > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$out/sq" -o p -- python3 "$root/tools/prof_sweep.py" 4096 12 bicgstab > /dev/null 2>&1
cd "$root"
python3 - "$out" <<'P' > "$out/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
def table(d):
    f = glob.glob(f"{out}/{d}/**/p_counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
for d in ("fetch", "write", "sq"):
    print("==", d)
    for k, cs in table(d).items():
        if "k_apply_vk" not in k: continue
        for c, v in cs.items():
            print(f"  {k:50s} {c:22s} launches {len(v):3d}  mean {sum(v) / len(v):16.1f}")
P
cat "$out/summary.txt"
