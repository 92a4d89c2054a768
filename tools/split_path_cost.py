#!/usr/bin/env python3
"""Cost of the multi-rank sweep schedule without a second GPU: rank 0 of a 2-rank strip with a transport that moves nothing
(ghost rows stay stale, so the coordinates are meaningless -- only the timing is).  Shows what the pack kernels, the
three-part K2x2 launch and the hook calls add to the single-rank pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
import torch
from turbomesh_amd import distributed as tmd
from turbomesh_amd.smoothing import solver

class NullHooks(tmd.HooksBase):
    def exchange(self, send, recv):
        pass
    def exchange_wait(self):
        pass
    def allreduce(self, t):
        pass

import ctypes as C
from turbomesh_amd import _capi
from turbomesh_amd.smoothing import smooth


class NativeNullHooks:
    """The same with the callbacks inside the library (what the RCCL transport costs on the host, minus RCCL)."""

    def __init__(self, mesh, owner, rank, world, option):
        self._owner = (C.c_int32 * len(owner))(*owner)
        self._hooks = _capi.tm_comm_hooks()
        _capi.check(_capi.lib().tm_debug_null_hooks(rank, world, self._owner, C.byref(self._hooks)))
        self.smoother = smooth.Smoother(mesh, option, None, hooks=self._hooks, stream=torch.cuda.current_stream().cuda_stream)

    def iterate(self, n):
        return self.smoother.iterate(n)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600   # timed behind ~0.25 s of untimed sweeps (settled clocks, see tools/dev/ramp_probe.py)
torch.cuda.set_device(0)
for world, cls in ((2, NullHooks), (3, NullHooks), (2, NativeNullHooks), (3, NativeNullHooks)):
    rank = 1 if world == 3 else 0      # world 3, rank 1: neighbours on both sides
    mesh = tmd.strip_for_rank(world, rank, n, n)
    h = cls(mesh, owner=list(range(world)), rank=rank, world=world, option=solver.Option.hip(inner=solver.Inner.relax))
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.25:
        h.iterate(120)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.iterate(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"world {world} rank {rank} {cls.__name__}: {dt / steps * 1e6:.1f} us per sweep ({n * n * steps / dt:.3e} nodes/s) with a null transport", flush=True)
    h.smoother.close()
