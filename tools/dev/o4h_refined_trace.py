#!/usr/bin/env python3
"""One Picard iteration of a refined O4H example with the plain solve (for a kernel trace).  usage: o4h_refined_trace.py [T106|LS89] [factor = 8] [inner = bicgstab]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "T106"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inner = solver.Inner[sys.argv[3]] if len(sys.argv) > 3 else solver.Inner.bicgstab
j = json.load(open(os.path.join(GOLD, "examples", name, name + ".json")))
nc = j["template"]["O4H"]["num_cells"]
for k in nc:
    nc[k] *= min(f, 2) if k == "o_grid" else f
inp = Input.parse(json.dumps(j))
mesh = inp.template.run(inp.geometry(GOLD))
with smooth.Smoother(mesh, solver.Option.hip(inner=inner, max_inner=200000), inp.wall_control_function) as sm:
    t0 = time.perf_counter()
    st = sm.iterate(1)
    dt = time.perf_counter() - t0
print(f"{name} x{f} {inner.name}: inner {st['inner_iterations']}, {dt * 1e3:.0f} ms, {dt * 1e6 / max(1, st['inner_iterations']):.1f} us per inner iteration", flush=True)
