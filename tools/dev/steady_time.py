#!/usr/bin/env python3
"""STEADY-STATE us per relaxation sweep of a lone n x n block: one handle, ~0.25 s of sweeps untimed (clock / power management settles
within ~50 ms, tools/dev/ramp_probe.py), then three timed calls.  usage: steady_time.py n|nixnj [...]   (TM_FUSE_3, TM_FUSE3_ROWS, TM_HIP_LIB; STEADY_SINGLE=1: one sweep per pass)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
for arg in sys.argv[1:] or ["4096"]:
    ni, nj = (int(v) for v in arg.split("x")) if "x" in arg else (int(arg), int(arg))
    mesh = configs.single_block(ni, nj)
    n = (ni * nj) ** 0.5
    k = max(600, int(0.05 / (55e-6 * (n / 4096.0) ** 2)))   # ~50 ms per call
    k -= k % 6
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=os.environ.get('STEADY_SINGLE', '0') != '0')) as sm:
        for _ in range(5):
            sm.iterate(k)
        out = [sm.iterate(k)["seconds"] / k * 1e6 for _ in range(3)]
    tag = f"FUSE_3={os.environ.get('TM_FUSE_3', 'default')} ROWS3={os.environ.get('TM_FUSE3_ROWS', 'default')}"
    print(f"{tag} {ni} x {nj}: " + " ".join(f"{o:.2f}" for o in out) + " us per sweep (steady state), "
          + f"{min(out) * 1e3 / (ni * nj) * 1e3:.4f} ps per node", flush=True)
