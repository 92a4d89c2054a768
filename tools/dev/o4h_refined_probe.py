#!/usr/bin/env python3
"""The O4H examples with every num_cells entry multiplied by a factor (every row kind of the reference -- interfaces, periodic
connections, 12 junction points, sliding inlet / outlet rows -- at 4, 9, 16 x the nodes), control function of the JSON or Laplace, two
Picard iterates of the GPU (default options, both inner solvers) against the sparse-LU oracle.  usage: o4h_refined_probe.py [T106|LS89] [factor ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "T106"
factors = [int(a) for a in sys.argv[2:]] or [2, 3]
rms = lambda p, q: float(np.sqrt(np.mean((p - q) ** 2)))
for f in factors:
    j = json.load(open(os.path.join(GOLD, "examples", name, name + ".json")))
    nc = j["template"]["O4H"]["num_cells"]
    for k in nc:
        nc[k] *= f
    inp = Input.parse(json.dumps(j))
    build = lambda: inp.template.run(inp.geometry(GOLD), tfi=None)
    mesh = build()
    nodes = sum(b.points.size[0] * b.points.size[1] for b in mesh.blocks)
    w = inp.wall_control_function.white
    for control, calgo in ((None, None), (("white", w.ds_target, w.theta_target), wcf.Algorithm(wcf.White(w.ds_target, w.theta_target)))):
        t = time.time()
        ref = oracle.picard_exact(OracleMesh(mesh), 2, control=control, keep_iterates=True)[1]
        ref = [np.concatenate([b.reshape(-1, 2) for b in it], axis=0) for it in ref]
        t_ref = time.time() - t
        print(f"{name} x{f}: {nodes} nodes, {'white' if control else 'laplace'}, oracle splu {t_ref:.1f} s", flush=True)
        for inner in (solver.Inner.bicgstab, solver.Inner.mg_bicgstab):
            m = build()
            out, inn = [], []
            t = time.time()
            with smooth.Smoother(m, solver.Option.hip(inner=inner, max_inner=400000), calgo) as sm:
                for k in range(2):
                    st = sm.iterate(1)
                    sm.download()
                    out.append(rms(mesh_flat(m), ref[k]))
                    inn.append((st["inner_iterations"], st["not_converged"]))
            print(f"   {inner.name:12s}: rms vs exact " + " ".join(f"{o:.2e}" for o in out) + f" | inner {inn} | {time.time() - t:.2f} s", flush=True)
