#!/usr/bin/env python3
"""K2x2 rows per chunk (tm_tune_fuse, measurement build) against time per relaxation sweep of a lone n x n block, several
repetitions per setting.  usage: relax_rows_probe.py n rows [rows ...]   (rows = 0: the library's own choice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))
from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1])
rows = [int(r) for r in sys.argv[2:]] or [0]
steps = 600 if n <= 2048 else 200
mesh = configs.single_block(n, n)
for r in rows:
    _capi.lib().tm_tune_fuse(r)
    out = []
    for rep in range(3):
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
            sm.iterate(40)
            st = sm.iterate(steps)
        out.append(st["seconds"] / steps * 1e6)
    print(f"{n}^2 rows {r:3d}: " + " ".join(f"{o:.2f}" for o in out) + " us per sweep", flush=True)
