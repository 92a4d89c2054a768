#!/usr/bin/env python3
"""One-off fuzz (not collected by pytest): K2x2 against single sweeps, bit for bit, on random block shapes, sweep counts and
relaxation factors -- exercises every combination of short / leftover row chunks and partial strip groups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(ncases):
    ni = int(rng.integers(5, 140)) if rng.random() < 0.6 else int(rng.integers(140, 900))
    nj = int(rng.integers(5, 300)) if rng.random() < 0.6 else int(rng.integers(300, 1500))
    sweeps = int(rng.integers(2, 9))
    omega = 1.0 if rng.random() < 0.5 else float(rng.uniform(0.5, 1.0))
    nb = int(rng.integers(1, 4))
    build = (lambda: configs.single_block(ni, nj, perturb=0.2)) if nb == 1 else (lambda: configs.strip(nb, ni, nj, reverse_odd=bool(rng.integers(0, 2))))
    out = []
    state = rng.bit_generator.state
    for single in (True, False):
        rng.bit_generator.state = state
        mesh = build()
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single, omega=omega)) as sm:
            sm.iterate(sweeps)
            sm.download()
        out.append(np.concatenate([b.points.data.reshape(-1, 2) for b in mesh.blocks]))
    ok = np.array_equal(out[0], out[1]) and np.isfinite(out[0]).all()
    bad += not ok
    print(f"case {case}: {nb} x {ni} x {nj}, {sweeps} sweeps, omega {omega:.3f}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
