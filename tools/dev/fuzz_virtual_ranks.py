#!/usr/bin/env python3
"""One-off fuzz (not collected by pytest): the multi-rank relaxation schedule (two streams, border / inside launches, direct or
packed sends) against the single-rank run, bit for bit, on random strips, owners and sweep counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import mesh_flat
from tests.test_gpu_virtual_ranks import _run_ranks, SplitThreadHooks
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for case in range(ncases):
    nb = int(rng.integers(2, 6))
    world = int(rng.integers(2, min(nb, 4) + 1))
    owner = [int(x) for x in rng.integers(0, world, nb)]
    for r in range(world):          # every rank owns something
        if r not in owner:
            owner[r % nb] = r
    if len(set(owner)) != world:
        continue
    ni, nj = int(rng.integers(5, 120)), int(rng.integers(5, 700))
    rev = bool(rng.integers(0, 2))
    sweeps = int(rng.integers(2, 11))
    omega = 1.0 if rng.random() < 0.5 else 0.9
    builder = lambda: configs.strip(nb, ni, nj, reverse_odd=rev)
    opt = solver.Option.hip(inner=solver.Inner.relax, omega=omega)
    ref = builder()
    smooth.mesh(ref, sweeps, opt)
    got = _run_ranks(builder, owner, opt, sweeps, hooks_cls=SplitThreadHooks if rng.random() < 0.5 else None)
    ok = np.array_equal(mesh_flat(got), mesh_flat(ref))
    bad += not ok
    print(f"case {case}: {nb} blocks {ni} x {nj} owners {owner} reversed {rev} sweeps {sweeps} omega {omega}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
