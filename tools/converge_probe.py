#!/usr/bin/env python3
"""Time to CONVERGED coordinates on the perturbed n^2 block: Picard + multigrid-preconditioned BiCGStab until the update of a Picard
iteration (smooth.zig:112-137) is <= 1e-10 RMS, for the default inner tolerance (exact Picard = the reference's semantics with an exact solver) and for
tolerances relative to each inner solve's initial residual (inexact Picard: same fixed point, fewer inner iterations).  usage: converge_probe.py [n = 4096] [rtol ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rtols = [float(a) for a in sys.argv[2:]] or [0.0, -1e-1, -1e-2, -1e-3, -1e-4]   # negative: relative to the INITIAL residual of each inner solve (TM_OPT_RTOL_INITIAL)
ref = None
for rtol in rtols:
    for rep in range(2):
        mesh = configs.single_block(n, n, perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=abs(rtol), rtol_initial=rtol < 0, check_every=1)) as sm:
            reached, st = sm.iterate_until_update(1e-10, 200)
            sm.download()
    xy = mesh.blocks[0].points.data
    if ref is None:
        ref = xy.copy()
    d = float(np.sqrt(np.mean((xy - ref) ** 2)))
    print(f"rtol {'default' if rtol == 0 else (str(-rtol) + ' of the initial residual' if rtol < 0 else rtol)}: reached {reached}, outer {st['outer_iterations']}, inner {st['inner_iterations']}, {st['seconds'] * 1e3:.1f} ms, "
          f"last update {np.sqrt((st['last_dx2'] + st['last_dy2']) / n / n):.1e}, rms from the default-tolerance fixed point {d:.1e}", flush=True)
