#!/usr/bin/env python3
"""Seam 1 as a Zig caller would use it: tm_smooth_mesh on HOST arrays (create + upload + iterate + download + destroy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mesh = configs.single_block(n, n, perturb=0.25)
smooth.mesh(mesh, 1, solver.Option.hip(inner=solver.Inner.relax))   # first call: module load etc.
for label, opt, its in (("relax, 200 sweeps", solver.Option.hip(inner=solver.Inner.relax), 200),
                        ("mg_bicgstab, 2 Picard iterations", solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-10), 2),
                        ("0 iterations (transfers + set-up only)", solver.Option.hip(inner=solver.Inner.relax), 0)):
    t0 = time.perf_counter()
    st = smooth.mesh(mesh, its, opt)
    dt = time.perf_counter() - t0
    print(f"tm_smooth_mesh {n}^2, {label}: {dt * 1e3:.1f} ms wall (device loop {st['seconds'] * 1e3:.1f} ms)", flush=True)
