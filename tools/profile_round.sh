#!/bin/bash
# Collects the profiles a round commits under profiles/ (run on the MI355X box through gpurun, from the repo root):
#   usage: tools/profile_round.sh r02
# kernel-trace + stats of the bench command (three sweeps per pass = the default on a block with fixed walls, two per pass, and --single-sweep), of the Krylov and multigrid paths, and the
# two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with tracing domains) that profiles/traffic.json is made from.
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name" -o p -- "$@" > "$out/$name.out" 2> "$out/$name.err"; cp "$out/$name"/p_kernel_stats.csv "$out/${tag}_${name}_kernel_stats.csv" 2>/dev/null;
  # the same run without the clock ramp of the cold device (launches starting within 60 ms of a kernel's first launch dropped): the file the bench line's `frac` is checked against
  python3 "$root/tools/steady_stats.py" "$out/$name"/p_kernel_trace.csv "$out/${tag}_${name}_kernel_steady.csv" --skip-ms 60 > "$out/$name.steady.txt" 2>&1;
  grep -h '^{' "$out/$name.out" > "$out/${tag}_${name}_line.json" 2>/dev/null; }
prof bench4096_k2x3 python3 "$root/bench.py" --no-cpu-baseline --no-solve
TM_FUSE_3=0 prof bench4096_k2x2 python3 "$root/bench.py" --no-cpu-baseline --no-solve
prof bench4096_k2_single python3 "$root/bench.py" --no-cpu-baseline --no-solve --single-sweep
prof bicgstab4096 python3 "$root/tools/prof_sweep.py" 4096 24 bicgstab
prof mg_solve4096 python3 "$root/tools/solve_probe.py" 4096 1e-6
prof config4_8x2048 python3 "$root/bench.py" --config 4 --no-cpu-baseline --steps 100
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python3 "$root/tools/prof_sweep.py" 4096 20 relax > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python3 "$root/tools/prof_sweep.py" 4096 20 relax > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$out/pmc_sq" -o p -- python3 "$root/tools/prof_sweep.py" 4096 21 relax > /dev/null 2>&1
cp "$out"/pmc_sq/p_counter_collection.csv "$out/${tag}_pmc_sq_k2x3_counter_collection.csv" 2>/dev/null
cp "$out"/pmc_fetch/p_counter_collection.csv "$out/${tag}_pmc_fetch_counter_collection.csv" 2>/dev/null
cp "$out"/pmc_write/p_counter_collection.csv "$out/${tag}_pmc_write_counter_collection.csv" 2>/dev/null
cd "$root"
python3 tools/pmc_traffic.py "$out/pmc_fetch" "$out/pmc_write" 4096 "$out/traffic.json" k_relax3 3 "$out/pmc_sq" > /dev/null
grep -h '^{' "$out/bench4096_k2x3.out" > "$out/${tag}_bench_n1_under_rocprof.json"
ls "$out"
