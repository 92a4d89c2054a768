#!/usr/bin/env python3
"""Time per BiCGStab iteration (plain, diagonal) on a perturbed n^2 block: one Picard solve capped at `iters` inner iterations.
usage: bicgstab_iter_probe.py [n = 4096] [iters = 200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mesh = configs.single_block(n, n, perturb=0.25)
for rep in range(2):
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-30, max_inner=iters, check_every=iters)) as sm:
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = sm.iterate(1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n}^2: {st['inner_iterations']} inner iterations in {dt * 1e3:.1f} ms = {dt / max(1, st['inner_iterations']) * 1e6:.1f} us per iteration "
          f"(TM_FUSE_P={os.environ.get('TM_FUSE_P', 'default')})", flush=True)
