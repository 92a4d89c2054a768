#!/usr/bin/env python3
"""us per sweep of consecutive iterate(k) calls on one handle from a cold start: does the figure depend on how long the GPU has been busy
(clock / power management)?  usage: ramp_probe.py [n = 4096] [sweeps per call = 200] [calls = 60]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 60
mesh = configs.single_block(n, n)
time.sleep(2.0)   # let the GPU fall back to its idle state
with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
    out = []
    t0 = time.perf_counter()
    for c in range(calls):
        st = sm.iterate(k)
        out.append((time.perf_counter() - t0, st["seconds"] / k * 1e6))
print(f"{n}^2, {k} sweeps per call, TM_FUSE_3={os.environ.get('TM_FUSE_3', 'default')}:")
print(" ".join(f"{t * 1e3:.0f}ms:{u:.1f}" for t, u in out))
