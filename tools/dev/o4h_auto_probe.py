#!/usr/bin/env python3
"""Does `inner = auto` choose well on REAL meshes?  The O4H examples with every num_cells entry multiplied by a factor (boundary-layer
clustering, cell aspect ratios of 10^3 and more), control function of the JSON, three Picard iterations with each inner strategy at the
default tolerance: inner iterations, ms, and what auto resolves to.  usage: o4h_auto_probe.py [T106|LS89] [factor ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "T106"
factors = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
for f in factors:
    j = json.load(open(os.path.join(GOLD, "examples", name, name + ".json")))
    nc = j["template"]["O4H"]["num_cells"]
    for k in nc:   # (the O-grid's wall-normal count stops at 2 x: its tanh clustering has a fixed first spacing, clustering.zig:56-95, which more cells than ~3 x would undercut)
        nc[k] *= min(f, 2) if k == "o_grid" else f
    inp = Input.parse(json.dumps(j))
    shapes = None
    def spread(xy):   # of log(g11 / g22) over the interior nodes: what a point smoother with block-mean semi-coarsening cannot follow
        a = xy[2:, 1:-1] - xy[:-2, 1:-1]
        b = xy[1:-1, 2:] - xy[1:-1, :-2]
        g11, g22 = (a ** 2).sum(-1), (b ** 2).sum(-1)
        ok = (g11 > 0) & (g22 > 0)
        l = np.log(g11[ok] / g22[ok])
        return float(l.std()), float(np.abs(l - l.mean()).max())
    m0 = inp.template.run(inp.geometry(GOLD))
    print(f"{name} x{f}: std / max deviation of log(g11/g22) per block: " + ", ".join(f"{b.points.size[0]}x{b.points.size[1]}: {spread(b.points.data)[0]:.2f} / {spread(b.points.data)[1]:.1f}" for b in m0.blocks), flush=True)
    for inner in (solver.Inner.auto, solver.Inner.bicgstab, solver.Inner.mg_bicgstab):
        mesh = inp.template.run(inp.geometry(GOLD))
        shapes = [tuple(b.points.size) for b in mesh.blocks]
        with smooth.Smoother(mesh, solver.Option.hip(inner=inner, max_inner=200000), inp.wall_control_function) as sm:
            t0 = time.perf_counter()
            st = sm.iterate(3)
            dt = time.perf_counter() - t0
            got = sm.inner
        print(f"{name} x{f} ({sum(a * b for a, b in shapes)} nodes, largest block {max(a * b for a, b in shapes)}) {inner.name:12s} -> {getattr(got, 'name', got)}: "
              f"inner {st['inner_iterations']:7d} over 3 Picard iterations, not converged {st['not_converged']}, {dt * 1e3:8.0f} ms", flush=True)
