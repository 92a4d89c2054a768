#!/usr/bin/env python3
"""What one GPU can measure of the multi-GPU scaling curves (there is no 8-GPU node on this pool): the time per sweep of the SLOWEST rank of
each partition -- a middle rank with neighbours on both sides, its blocks, its chain, its announce-and-wait kernels -- with a transport that
moves nothing but occupies the chain's queue for a set time per exchange (TM_NULL_EXCHANGE_US: the device time of a real RCCL exchange, the
one term only an 8-GPU run can measure), against the measured single-GPU run of the whole job.

  configs[3] (strong scaling): 8 coupled 2048^2 blocks over N = 1, 2, 4, 8 GPUs (8 / N blocks per rank)
  default line (weak scaling): N coupled 4096^2 blocks, one per GPU

A PROJECTION from measured components, not a measurement of scaling: printed as such.   usage: scaling_projection.py [exchange_us ...]"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ["TM_ROOT"])
os.environ.setdefault("TM_HIP_LIB", os.path.join(os.environ["TM_ROOT"], "turbomesh_amd", "libtm_hip_dbg.so"))
import torch
from turbomesh_amd import _capi, configs, distributed as tmd
from turbomesh_amd.smoothing import smooth, solver
world, bpr, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
opt = solver.Option.hip(inner=solver.Inner.relax)
torch.cuda.set_device(0)
if world == 1:
    sm = smooth.Smoother(configs.strip(bpr, n, n) if bpr > 1 else configs.single_block(n, n), opt)
else:
    rank = world // 2 if world > 2 else 0     # a rank with neighbours on both sides when there is one
    nb = world * bpr
    owner = (C.c_int32 * nb)(*[b // bpr for b in range(nb)])
    hooks = _capi.tm_comm_hooks()
    _capi.check(_capi.lib().tm_debug_null_hooks(rank, world, owner, C.byref(hooks)))
    sm = smooth.Smoother(tmd.strip_for_rank(world, rank, n, n, blocks_per_rank=bpr), opt, None, hooks=hooks, stream=torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    sm.iterate(120); torch.cuda.synchronize()
best = 1e30
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); sm.iterate(300); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
print(best)
"""


def run(world, bpr, n, us):
    env = dict(os.environ, TM_ROOT=ROOT, TM_NULL_EXCHANGE_US=str(us))
    r = subprocess.run([sys.executable, "-c", CHILD, str(world), str(bpr), str(n)], capture_output=True, text=True, env=env)
    if r.returncode:
        raise SystemExit(r.stderr[-2000:])
    return float(r.stdout.strip().splitlines()[-1])


def main():
    delays = [float(a) for a in sys.argv[1:]] or [0.0, 10.0, 20.0]
    print("# PROJECTION from components measured on ONE MI355X (tools/scaling_projection.py): us per sweep of the slowest rank, null transport whose")
    print("# exchange occupies the chain's queue for the stated time; speed-up = the measured 1-GPU run of the whole job / that.  Not a scaling measurement.")
    t1 = run(1, 8, 2048, 0)
    print(f"configs[3], 8 x 2048^2 coupled blocks, N = 1 (measured, all blocks in one process): {t1:.1f} us per sweep")
    for us in delays:
        row = []
        for world in (2, 4, 8):
            t = run(world, 8 // world, 2048, us)
            row.append(f"N = {world}: {t:5.1f} us ({t1 / t:4.2f} x)")
        print(f"  exchange {us:4.0f} us per triple on the chain -> " + "   ".join(row))
    w1 = run(1, 1, 4096, 0)
    print(f"default line, 4096^2 per GPU, N = 1 (measured, lone block with fixed walls): {w1:.1f} us per sweep")
    for us in delays:
        t = run(3, 1, 4096, us)
        print(f"  exchange {us:4.0f} us per triple on the chain -> a middle rank of the strip: {t:5.1f} us per sweep = weak-scaling efficiency {w1 / t:4.2f} at any N > 1")


if __name__ == "__main__":
    main()
