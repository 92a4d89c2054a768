#!/usr/bin/env python3
"""us per relaxation sweep (two per pass) of a lone n x n block, three repetitions.  usage: sweep_time.py n [n ...]   (TM_HIP_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
for n in [int(a) for a in sys.argv[1:]] or [2048]:
    steps = 800 if n <= 2048 else 200
    mesh = configs.single_block(n, n)
    out = []
    for rep in range(3):
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
            sm.iterate(40)
            st = sm.iterate(steps)
        out.append(st["seconds"] / steps * 1e6)
    print(f"{os.path.basename(os.environ.get('TM_HIP_LIB', 'libtm_hip.so'))} {n}^2: " + " ".join(f"{o:.2f}" for o in out) + " us per sweep", flush=True)
