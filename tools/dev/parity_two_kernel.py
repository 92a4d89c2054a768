#!/usr/bin/env python3
"""Which BiCGStab recurrence lands closer to the exact Picard iterate?  strip(2, 200, 700) and a perturbed 300 x 500 block,
3 Picard iterations, rtol 1e-13: classic (TM_FUSE_2=0), two-kernel (default), eager.  Run from the repo root on the GPU box.
(Uses the test-side oracle as the checker, like the tests do.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

cases = {"strip2_big": lambda: configs.strip(2, 200, 700, reverse_odd=True), "block300x500": lambda: configs.single_block(300, 500, perturb=0.25)}
for name, build in cases.items():
    om = OracleMesh(build())
    oracle.picard_exact(om, 3)
    exact = om.flat()
    for label, env, eager in (("classic", "0", False), ("two-kernel", "1", False), ("eager", "1", True)):
        os.environ["TM_FUSE_2"] = env
        mesh = build()
        with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-13, max_inner=20000, eager_scalars=eager)) as sm:
            st = sm.iterate(3)
            sm.download()
        rms = float(np.sqrt(np.mean((mesh_flat(mesh) - exact) ** 2)))
        print(f"{name:14s} {label:18s}: rms vs exact {rms:.3e}, inner iterations {st['inner_iterations']}, not converged {st['not_converged']}", flush=True)
