#!/usr/bin/env python3
"""For rocprofv3 --kernel-trace: a short run of the multi-rank sweep schedule (rank 1 of 3, null transport inside the library)
so that the kernel timeline of one sweep pair (inside pass, border pass, perimeter rows, gaps) can be read off the trace.
usage: split_path_trace.py [n = 2048] [sweeps = 40]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
import torch

from turbomesh_amd import _capi
from turbomesh_amd import distributed as tmd
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
world, rank = 3, 1
torch.cuda.set_device(0)
mesh = tmd.strip_for_rank(world, rank, n, n)
owner = (C.c_int32 * world)(*range(world))
hooks = _capi.tm_comm_hooks()
_capi.check(_capi.lib().tm_debug_null_hooks(rank, world, owner, C.byref(hooks)))
sm = smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax), None, hooks=hooks, stream=torch.cuda.current_stream().cuda_stream)
sm.iterate(8)
torch.cuda.synchronize()
sm.iterate(steps)
torch.cuda.synchronize()
sm.close()
print("done")
