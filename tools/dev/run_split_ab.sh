# multi-rank / coupled sweep triples: correctness of the schedule, then one rank of a strip with a transport that moves nothing
timeout -k 10 600 python -m pytest tests/test_gpu_virtual_ranks.py tests/test_gpu_round3_api.py tests/test_bench_contract.py -x -q 2>&1 | tail -2
for n in 4096 2048 1024; do python3 tools/split_path_cost.py $n 2>/dev/null | grep Native; done
python3 tools/config4_probe.py 2>/dev/null | head -1
