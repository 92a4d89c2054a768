#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: N relax sweeps of the 4096^2 block through the C-ABI handle (no CPU baseline, no events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "relax"
single = len(sys.argv) > 4 and sys.argv[4] == "single"   # one sweep per kernel pass (K2) instead of pairs (K2x2)
if os.environ.get('TM_FUSE_ROWS'):
    _capi.lib().tm_tune_fuse(int(os.environ['TM_FUSE_ROWS']))
mesh = configs.single_block(n, n)
if mode == "relax":
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)) as sm:
        sm.iterate(steps)
else:
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.bicgstab, max_inner=steps, rtol=1e-30)) as sm:
        sm.iterate(1)
print("done")
