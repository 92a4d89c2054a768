#!/bin/bash
# A/B of two builds (tools/dev/lib_old.so, lib_new.so) on ONE box: bench launch time and the multi-rank schedule of one rank
for v in old new old new; do
  cp tools/dev/lib_$v.so turbomesh_amd/libtm_hip.so
  python bench.py --no-cpu-baseline --no-solve 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', 'ms_per_step', round(d['ms_per_step']*1e3,2), 'launch_us', round(d['roofline']['avg_launch_us'],2))"
  for n in 4096 2048; do echo $v $n $(python tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | cut -c1-70); done
done
