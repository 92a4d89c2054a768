#!/usr/bin/env python3
"""One-off soak (not collected by pytest): repeated long runs of the sweep kernels must agree bit for bit with each other and with
single sweeps -- the way the buffer_store / SGPR-offset hazard showed up (sporadic 64-byte groups) before it was understood."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ref = None
bad = 0
for run, single in enumerate((False, False, True, False, True)):
    mesh = configs.single_block(n, n, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)) as sm:
        sm.iterate(sweeps)
        sm.download()
    d = mesh.blocks[0].points.data
    if ref is None:
        ref = d.copy()
    same = np.array_equal(d, ref)
    bad += not same
    print(f"run {run} single_sweep={single}: {'identical' if same else 'DIFFERENT (%d nodes)' % int(np.any(d != ref, axis=2).sum())}", flush=True)
sys.exit(1 if bad else 0)
