#!/usr/bin/env python3
"""Inner-iteration counts of the two Picard solvers on the reference's example meshes (boundary-layer clustering)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("T106", "LS89"):
    for control in ("laplace", "white"):
        for inner in (solver.Inner.mg_bicgstab, solver.Inner.bicgstab):
            inp = Input.parse(open(os.path.join(GOLD, "examples", name, name + ".json")).read())
            mesh = inp.template.run(inp.geometry(GOLD))
            algo = inp.wall_control_function if control == "white" else wcf.Algorithm.laplace()
            with smooth.Smoother(mesh, solver.Option.hip(inner=inner, rtol=1e-10, max_inner=20000, check_every=1), algo) as sm:
                t0 = time.perf_counter()
                st = sm.iterate(3)
                dt = time.perf_counter() - t0
            print(f"{name} {control:8s} {inner.name:12s}: inner {st['inner_iterations']:6d} over 3 Picard iterations, not_conv {st['not_converged']}, {dt*1e3:.0f} ms", flush=True)
