#!/usr/bin/env python3
"""Time to a scaled nonlinear residual <= 1e-8 on the perturbed n^2 block for several inner tolerances (inexact Picard)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rtols = [float(a) for a in sys.argv[2:]] or [1e-10, 1e-6, 1e-4, 1e-3, 1e-2]
for rtol in rtols:
    mesh = configs.single_block(n, n, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=rtol)) as sm:
        reached, st = sm.iterate_until(1e-8, 50)
    print(f"rtol {rtol:g}: reached {reached}, outer {st['outer_iterations']}, inner {st['inner_iterations']}, {st['seconds'] * 1e3:.1f} ms, residual {st['scaled_residual_rms']:.2e}", flush=True)
