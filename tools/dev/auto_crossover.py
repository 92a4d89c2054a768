#!/usr/bin/env python3
"""Where does the multigrid-preconditioned solve overtake the plain one on the synthetic uniform blocks (the size rule of TM_INNER_AUTO)?
Perturbed n^2 block, three Picard iterations at the default tolerance with each, ms and inner iterations.  usage: auto_crossover.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
cases = [("%d^2" % n, (lambda n=n: configs.single_block(n, n, perturb=0.25))) for n in ([int(a) for a in sys.argv[1:]] or [200, 256, 317, 400, 512, 724, 1024])]
if len(sys.argv) <= 1 or os.environ.get("CROSSOVER_STRIPS"):
    def strip(k, ni, nj):
        import numpy as np
        m = configs.strip(k, ni, nj)
        rng = np.random.default_rng(5)
        for b in m.blocks:
            d = b.points.data
            d[1:-1, 1:-1] += 0.25 / d.shape[0] * (rng.random(d[1:-1, 1:-1].shape) - 0.5) * (d[-1, 0, 1] - d[0, 0, 1])
        return m
    cases += [(f"strip {k} x {ni} x {nj}", (lambda k=k, ni=ni, nj=nj: strip(k, ni, nj))) for k, ni, nj in ([tuple(int(x) for x in t.split('x')) for t in os.environ['CROSSOVER_STRIPS'].split(',')] if 'x' in os.environ.get('CROSSOVER_STRIPS', '') else ((4, 64, 256), (8, 64, 64), (8, 128, 128), (2, 200, 200), (8, 256, 256)))]
for name, build in cases:
    row = []
    for inner in ((solver.Inner.mg_bicgstab, solver.Inner.mg_bicgstab) if os.environ.get("CROSSOVER_MG_ONLY") else (solver.Inner.bicgstab, solver.Inner.mg_bicgstab)):
        best = None
        for rep in range(2):
            mesh = build()
            with smooth.Smoother(mesh, solver.Option.hip(inner=inner)) as sm:
                t0 = time.perf_counter()
                st = sm.iterate(3)
                dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row.append((best, st["inner_iterations"], st["not_converged"]))
    print(f"{name:18s}  bicgstab {row[0][0] * 1e3:8.1f} ms ({row[0][1]} inner, not converged {row[0][2]})   mg_bicgstab {row[1][0] * 1e3:8.1f} ms ({row[1][1]} inner)   ratio {row[1][0] / row[0][0]:.2f}", flush=True)
