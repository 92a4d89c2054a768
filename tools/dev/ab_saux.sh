#!/bin/bash
# A/B of the cache policy of K2x2's result stores (-DTM_R2_SAUX=n builds under tools/dev/ab/) on ONE box: launch time of the bench line
for rep in 1 2; do
for v in default 0 1 3 18; do
  lib=turbomesh_amd/libtm_hip.so; [ $v != default ] && lib=tools/dev/ab/libtm_saux$v.so
  TM_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-solve 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('saux $v', 'us_per_sweep', round(d['ms_per_step']*1e3,2), 'launch_us', round(d['roofline']['avg_launch_us'],2), 'stream', d['roofline'].get('stream'))"
done; done
