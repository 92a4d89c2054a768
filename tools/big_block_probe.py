#!/usr/bin/env python3
"""One-off: a 16384^2 block (268 M nodes, 4 GiB per vector) through TFI + relaxation pairs + one multigrid-preconditioned solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
t0 = time.perf_counter(); mesh = configs.single_block(n, n); print(f"TFI + host copy of {n}^2: {time.perf_counter() - t0:.1f} s", flush=True)
d = mesh.blocks[0].points.data
edges = [d[0].copy(), d[-1].copy(), d[:, 0].copy(), d[:, -1].copy()]
with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
    sm.iterate(4)
    torch.cuda.synchronize(); t0 = time.perf_counter(); st = sm.iterate(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"relax: {dt / 40 * 1e6:.0f} us per sweep = {n * n * 40 / dt:.3e} nodes/s, residual {st['last_residual']:.3e}", flush=True)
    sm.download()
assert np.isfinite(d).all()
assert all(np.array_equal(a, b) for a, b in zip(edges, [d[0], d[-1], d[:, 0], d[:, -1]]))
with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-10)) as sm:
    st = sm.iterate(1)
    print(f"mg_bicgstab: 1 Picard solve, {st['inner_iterations']} inner iterations, {st['seconds'] * 1e3:.0f} ms, scaled residual at start {st['scaled_residual_rms']:.3e}", flush=True)
print("ok")
