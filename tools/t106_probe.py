#!/usr/bin/env python3
"""BASELINE configs[0] / [2] on the GPU: examples/<name>/<name>.json as written (10 Picard iterations, white control function) through
the hip solver; prints wall time and inner iterations.  usage: t106_probe.py [T106|LS89] [rtol = 1e-8 | gmres]   (gmres: GMRES(30) on the device, rtol 1e-8)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TM_HIP_LIB", os.path.join(ROOT, "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

name = sys.argv[1] if len(sys.argv) > 1 else "T106"
use_gmres = len(sys.argv) > 2 and sys.argv[2] == "gmres"
rtol = float(sys.argv[2]) if len(sys.argv) > 2 and not use_gmres else 1e-8
gold = os.path.join(ROOT, "tests", "golden")
inp = Input.parse(open(os.path.join(gold, "examples", name, name + ".json")).read())
mesh = inp.template.run(inp.geometry(gold), tfi=None)
w = inp.wall_control_function.white
if os.environ.get("TM_ROWS"):   # rows per K2 chunk (tm_tune_apply), for experiments
    from turbomesh_amd import _capi
    _capi.lib().tm_tune_apply(int(os.environ["TM_ROWS"]), 0, -1, -1)
for rep in range(2):
    opt = solver.Option.hip(inner=solver.Inner.gmres, rtol=rtol, max_inner=100000) if use_gmres else solver.Option.hip(rtol=rtol, max_inner=20000)
    with smooth.Smoother(mesh, opt, wcf.Algorithm(wcf.White(w.ds_target, w.theta_target))) as sm:
        t0 = time.perf_counter()
        st = sm.iterate(inp.iterations)
        dt = time.perf_counter() - t0
    print(f"{name}{' [gmres(30), diagonal]' if use_gmres else ''}: {inp.iterations} Picard iterations, {st['inner_iterations']} inner iterations, {dt * 1e3:.1f} ms "
          f"({dt / max(1, st['inner_iterations']) * 1e6:.1f} us per inner iteration), not converged {st['not_converged']}", flush=True)
