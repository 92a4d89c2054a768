#!/usr/bin/env python3
"""Seam 2 on the reference's own example: the oracle assembles the T106 system (iteration 0, Laplace), tm_csr_solve solves both
components; wall time per call (PCIe inclusive) and per inner iteration, lazy scalar steps vs a launch per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from tests.test_o4h import load
from tests.test_gpu_csr_solve import csr_solve
name = sys.argv[1] if len(sys.argv) > 1 else "T106"
_, mesh = load(name, oracle_tfi)
sy = oracle.System(OracleMesh(mesh))
sy.fill(0); sy.fill_x_specific()
Ap, Ai, Ax, bx = sy.lhs_p.copy(), sy.lhs_i.copy(), sy.lhs_values.copy(), sy.rhs_x.copy()
sy.fill_y_specific()
Ay, by = sy.lhs_values.copy(), sy.rhs_y.copy()
sy.seed_initial_guess()
x0, y0 = sy.x_new.copy(), sy.y_new.copy()
for eager in (False, True, False, True):
    t0 = time.perf_counter()
    x, y, rc, st = csr_solve(Ap, Ai, Ax, bx, by, x0=x0, y0=y0, Ay=Ay, rtol=1e-10, max_inner=20000, eager_scalars=eager)
    dt = time.perf_counter() - t0
    print(f"{name} n={len(bx)} {'launch per step' if eager else 'lazy steps     '}: rc {rc}, {st['inner_iterations']} inner iterations, {dt * 1e3:.1f} ms wall, "
          f"{st['seconds'] * 1e6 / max(1, st['inner_iterations']):.1f} us per iteration (whole call)", flush=True)
# ILU(0) in the slot (TM_OPT_PRECOND_ILU0): the reference's preconditioner of its example inputs, level-scheduled on the device
from turbomesh_amd.smoothing import solver
for rep in range(2):
    t0 = time.perf_counter()
    x, y, rc, st = csr_solve(Ap, Ai, Ax, bx, by, x0=x0, y0=y0, Ay=Ay, rtol=1e-10, max_inner=5000, preconditioner=solver.Preconditioner.ilu0)
    dt = time.perf_counter() - t0
    print(f"{name} n={len(bx)} ILU(0), level-scheduled: rc {rc}, {st['inner_iterations']} inner iterations, {dt * 1e3:.1f} ms wall, "
          f"{st['seconds'] * 1e6 / max(1, st['inner_iterations']):.0f} us per iteration (whole call, analysis + factorisation included)", flush=True)
# the CPU oracle's faithful BiCGStab + ILU(0) on the same system, x component, for the iteration count (its stop test is the reference's unscaled one)
t0 = time.perf_counter()
xo, its, ok = oracle.csr_solve("bicgstab", len(bx), Ap, Ai, Ax, bx, x0=x0, precond=oracle.PRECOND_ILU0, max_iters=5000, rtol=1e-12, atol=0.0)
print(f"{name} CPU oracle BiCGStab + ILU(0), x component, rtol 1e-12 (unscaled): {its} iterations, {(time.perf_counter() - t0) * 1e3:.1f} ms, converged {ok}")
