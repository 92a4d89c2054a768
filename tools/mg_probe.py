#!/usr/bin/env python3
"""Convergence / cost of the Picard inner solve with and without the multigrid preconditioner."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
outer = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-10
res = {}
for name, inner in (("mg_bicgstab", solver.Inner.mg_bicgstab), ("bicgstab", solver.Inner.bicgstab)):
    if name == "bicgstab" and n > 2048 and os.environ.get("MG_ONLY"):
        continue
    mesh = configs.single_block(n, n, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=inner, rtol=rtol, max_inner=20000, check_every=4)) as sm:
        for k in range(outer):
            t0 = time.perf_counter()
            st = sm.iterate(1)
            dt = time.perf_counter() - t0
            print(f"{name:12s} n={n} outer {k}: inner {st['inner_iterations']:6d} sweeps {st['operator_sweeps']:6d} not_conv {st['not_converged']} "
                  f"scaled_res_rms {st['scaled_residual_rms']:.3e} residual {st['last_residual']:.3e} time {dt*1e3:.1f} ms", flush=True)
        sm.download()
    res[name] = mesh.blocks[0].points.data.copy()
if len(res) == 2:
    d = res["mg_bicgstab"] - res["bicgstab"]
    print("rms difference between the two solvers' iterates:", float(np.sqrt(np.mean(d * d))))
