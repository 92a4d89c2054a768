# tests of the K2x3 paths, then the bench workloads that use them (usage on the GPU box: bash tools/dev/run_cfg45.sh)
python -m pytest tests/test_gpu_round3_api.py tests/test_gpu_benchsize.py tests/test_gpu_fused.py tests/test_gpu_virtual_ranks.py -x -q 2>&1 | tail -2
python tools/dev/steady_time.py 4096 2048 1448 1024 512 3000x5000
TM_FUSE3_ROWS=50 python tools/dev/steady_time.py 4096
for r in 0; do
TM_FUSE3_ROWS=$r python bench.py --config 4 --no-cpu-baseline > gpurun_out/config4_r$r.json 2>/dev/null
TM_FUSE3_ROWS=$r python bench.py --config 5 --no-cpu-baseline > gpurun_out/config5_r$r.json 2>/dev/null
done
python - <<'PY'
import json
for c in (4, 5):
    for r in (0,):
        f = f"gpurun_out/config{c}_r{r}.json"
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
for n in 4096 2048; do python3 tools/split_path_cost.py $n 2>/dev/null | grep Native; done
