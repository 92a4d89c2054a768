#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU's share: 8 of the 64 independent 2048^2 slices, relaxation sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
nsl, n = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mesh = configs.slices(nsl, n)
with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.25:   # settled clocks (tools/dev/ramp_probe.py)
        sm.iterate(120)
        torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter(); sm.iterate(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{nsl} independent slices of {n}^2 on one GPU: {dt / 200 * 1e6:.1f} us per sweep of all slices, {nsl * n * n * 200 / dt:.3e} nodes/s")
