#!/bin/bash
# A/B of two builds of libtm_hip.so on ONE box: tools/dev/lib_old.so vs tools/dev/lib_new.so, bench line launch time (K2x2 and K2)
for v in old new old new; do
  cp tools/dev/lib_$v.so turbomesh_amd/libtm_hip.so
  for flag in "" "--single-sweep"; do
    python bench.py --no-cpu-baseline --no-solve $flag 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', '$flag', 'ms_per_step', round(d['ms_per_step']*1e3,2), 'launch_us', round(d['roofline']['avg_launch_us'],2))"
  done
done
