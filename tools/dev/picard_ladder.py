#!/usr/bin/env python3
"""Size ladder of Picard parity: a perturbed n x n block (and a 2 x n strip), two Picard iterates through the GPU's `bicgstab` and
`mg_bicgstab` with the library's DEFAULT options against the exact-solve oracle (scipy splu of the oracle-assembled CSR =
the reference with its UMFPACK backend, umfpack.zig:18-24), next to the oracle's distance from ITSELF with another elimination
order (the fp64 floor of "the exact iterate").  usage: picard_ladder.py [sizes...] [--floor-max N] [--rtol R]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="*", type=int, default=[129, 257, 513, 1025])
ap.add_argument("--floor-max", type=int, default=1025, help="largest n for the second elimination order")
ap.add_argument("--band-max", type=int, default=513, help="largest n for the oracle's banded LU")
ap.add_argument("--rtol", type=float, nargs="*", default=[0.0])
ap.add_argument("--strip", action="store_true")
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--no-oracle", action="store_true", help="sizes the sparse LU cannot hold: the GPU's tightest plain-BiCGStab run (rtol 1e-16) stands in as the reference")
a = ap.parse_args()
rms = lambda p, q: float(np.sqrt(np.mean((p - q) ** 2)))

import threading
def _heartbeat():   # a sparse LU of 4 M unknowns says nothing for minutes; gpurun takes silence for a hang
    t0 = time.time()
    while True:
        time.sleep(60)
        print(f"   ... {time.time() - t0:.0f} s", flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()


def flat(its):
    return [np.concatenate([b.reshape(-1, 2) for b in it], axis=0) for it in its]


for n in a.sizes:
    build = (lambda: configs.strip(2, n, n)) if a.strip else (lambda: configs.single_block(n, n, perturb=0.25))
    if a.strip:   # roughen the strip the same way (the interface stays matched: interior nodes only)
        def build(n=n):
            m = configs.strip(2, n, n)
            rng = np.random.default_rng(7)
            for b in m.blocks:
                d = b.points.data
                d[1:-1, 1:-1] += 0.25 / (n - 1) * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
            return m
    seed = build()
    t = time.time()
    if a.no_oracle:
        ref = []
        m = build()
        with smooth.Smoother(m, solver.Option.hip(rtol=1e-16, max_inner=400000)) as sm:
            for k in range(a.iters):
                st = sm.iterate(1)
                sm.download()
                ref.append(mesh_flat(m).copy())
                print(f"   reference = GPU bicgstab rtol 1e-16: iterate {k + 1}: inner {st['inner_iterations']} not_converged {st['not_converged']}", flush=True)
    else:
        ref = flat(oracle.picard_exact(OracleMesh(seed), a.iters, keep_iterates=True)[1])
    t_ref = time.time() - t
    line = f"n {n}{' strip2' if a.strip else ''}: {'gpu rtol 1e-16' if a.no_oracle else 'oracle splu'} {t_ref:.1f} s |"
    if n <= a.floor_max and not a.no_oracle:
        alt = flat(oracle.picard_exact(OracleMesh(seed), a.iters, keep_iterates=True, permc_spec="MMD_AT_PLUS_A")[1])
        line += " floor(COLAMD vs MMD) " + " ".join(f"{rms(x, y):.1e}" for x, y in zip(ref, alt)) + " |"
    if n <= a.band_max and not a.no_oracle:
        bnd = flat(oracle.picard_direct(OracleMesh(seed), a.iters, keep_iterates=True)[1])
        line += " floor(splu vs banded) " + " ".join(f"{rms(x, y):.1e}" for x, y in zip(ref, bnd)) + " |"
    print(line, flush=True)
    step = rms(ref[0], mesh_flat(seed))
    for inner in (solver.Inner.bicgstab, solver.Inner.mg_bicgstab):
        for rtol in a.rtol:
            mesh = build()
            out, inn = [], []
            t = time.time()
            with smooth.Smoother(mesh, solver.Option.hip(inner=inner, rtol=rtol, max_inner=200000)) as sm:
                for k in range(a.iters):
                    st = sm.iterate(1)
                    sm.download()
                    out.append(rms(mesh_flat(mesh), ref[k]))
                    inn.append((st["inner_iterations"], st["not_converged"]))
            print(f"   {inner.name:12s} rtol {rtol or 1e-14:g}: rms vs exact " + " ".join(f"{o:.2e}" for o in out) + f" | inner {inn} | {time.time() - t:.2f} s"
                  f" | first step moves {step:.1e}", flush=True)
