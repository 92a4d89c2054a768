#!/usr/bin/env python3
"""One-off soak (not collected by pytest): three sweeps per pass on COUPLED blocks -- single process (blocks of >= 2^20 owned nodes)
and virtual ranks (depth-3 halo, counter joins between the two queues) -- against one sweep per pass, bit for bit, on random shapes,
owners and sweep counts.  usage: soak_triples.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import mesh_flat
from tests.test_gpu_virtual_ranks import _run_ranks, SplitThreadHooks
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
single = solver.Option.hip(inner=solver.Inner.relax, single_sweep=True)
fused = solver.Option.hip(inner=solver.Inner.relax)

# ---- one process, coupled blocks of a million nodes and more
builders = [
    ("strip 4 x (520 x 610) reversed", lambda: configs.strip(4, 520, 610, reverse_odd=True)),
    ("two_by_two 600 x 540", lambda: configs.two_by_two(600, 540)),
    ("periodic channel 1100 x 1000, sliding", lambda: configs.periodic_channel(1100, 1000)),
    ("strip 2 x (1024 x 1024)", lambda: configs.strip(2, 1024, 1024)),
]
for name, build in builders:
    for sweeps in (int(rng.integers(3, 12)), int(rng.integers(30, 50))):
        ref = build()
        smooth.mesh(ref, sweeps, single)
        for rep in range(2):
            got = build()
            smooth.mesh(got, sweeps, fused)
            ok = np.array_equal(mesh_flat(got), mesh_flat(ref))
            bad += not ok
            print(f"one process, {name}, {sweeps} sweeps, run {rep}: {'ok' if ok else 'MISMATCH'}", flush=True)

# ---- virtual ranks, triples forced on small blocks as well
os.environ["TM_TRIPLES_MIN_NODES"] = "1"
for case in range(ncases):
    nb = int(rng.integers(2, 6))
    world = int(rng.integers(2, min(nb, 4) + 1))
    owner = [int(x) for x in rng.integers(0, world, nb)]
    for r in range(world):
        if r not in owner:
            owner[r % nb] = r
    if len(set(owner)) != world:
        continue
    big = rng.random() < 0.3
    ni, nj = (int(rng.integers(200, 500)), int(rng.integers(600, 1500))) if big else (int(rng.integers(16, 120)), int(rng.integers(16, 700)))
    rev = bool(rng.integers(0, 2))
    sweeps = int(rng.integers(3, 40))
    builder = lambda: configs.strip(nb, ni, nj, reverse_odd=rev)
    ref = builder()
    smooth.mesh(ref, sweeps, single)
    got = _run_ranks(builder, owner, fused, sweeps, hooks_cls=SplitThreadHooks if rng.random() < 0.5 else None)
    ok = np.array_equal(mesh_flat(got), mesh_flat(ref))
    bad += not ok
    print(f"ranks, case {case}: {nb} blocks {ni} x {nj} owners {owner} reversed {rev} sweeps {sweeps}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
