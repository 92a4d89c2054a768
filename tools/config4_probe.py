#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: 8 blocks of 2048^2 coupled by 7 interfaces, relaxation sweeps (nodes/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
nb, n = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mesh = configs.strip(nb, n, n)
for single in (False, True):
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)) as sm:
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 0.25:   # settled clocks (tools/dev/ramp_probe.py)
            sm.iterate(120)
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = sm.iterate(200)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{nb} x {n}^2 on one GPU, single_sweep={single}: {dt / 200 * 1e6:.1f} us per sweep, {nb * n * n * 200 / dt:.3e} nodes/s", flush=True)
