#!/usr/bin/env python3
"""Random sizes of every parametric topology (strips plain / reversed, periodic channels sliding / fixed, 2 x 2 with a junction, the
plate with the White control function) through the default single-process solver: (a) 2 Picard iterations against the exact-solve
oracle, (b) five capped iterations of the fused two-kernel BiCGStab against the launch-per-step path (indexing of every row kind in
the fused perimeter-row code: connected, junction, sliding, periodic, interface).  usage: fuzz_picard_topologies.py [cases = 40] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat, oracle_tfi
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
worst_exact = worst_pair = 0.0
for k in range(cases):
    kind = k % 6
    ni, nj = int(rng.integers(7, 60)), int(rng.integers(7, 150))
    control = None
    if kind == 0:
        nb = int(rng.integers(2, 12))
        name, build = f"strip {nb} x {ni} x {nj}", (lambda tfi=None, nb=nb, ni=ni, nj=nj: configs.strip(nb, ni, nj, tfi=tfi))
    elif kind == 1:
        nb = int(rng.integers(2, 6))
        name, build = f"strip reversed {nb} x {ni} x {nj}", (lambda tfi=None, nb=nb, ni=ni, nj=nj: configs.strip(nb, ni, nj, tfi=tfi, reverse_odd=True))
    elif kind == 2:
        name, build = f"channel sliding {ni} x {nj}", (lambda tfi=None, ni=ni, nj=nj: configs.periodic_channel(ni, nj, tfi=tfi))
    elif kind == 3:
        name, build = f"channel fixed {ni} x {nj}", (lambda tfi=None, ni=ni, nj=nj: configs.periodic_channel(ni, nj, tfi=tfi, sliding=False))
    elif kind == 4:
        n, m = int(rng.integers(6, 40)), int(rng.integers(6, 60))
        name, build = f"two_by_two {n} x {m}", (lambda tfi=None, n=n, m=m: configs.two_by_two(n, m, tfi=tfi))
    else:
        ni, nj = int(rng.integers(9, 50)), int(rng.integers(7, 30))
        name, build = f"plate white {ni} x {nj}", (lambda tfi=None, ni=ni, nj=nj: configs.plate(ni, nj, tfi=tfi))
        control = wcf.Algorithm(wcf.White(0.02))
    om = OracleMesh(build(oracle_tfi))
    oracle.picard_exact(om, 2, control=("white", 0.02, np.pi / 2) if control else None)
    mesh = build()
    inner = solver.Inner[os.environ.get("FUZZ_INNER", "bicgstab")]   # FUZZ_INNER=mg_bicgstab: the multigrid-preconditioned solve (its perimeter treatment on every row kind)
    with smooth.Smoother(mesh, solver.Option.hip(inner=inner), control) as sm:   # the library's defaults
        st = sm.iterate(2)
        sm.download()
    rms = float(np.sqrt(np.mean((mesh_flat(mesh) - om.flat()) ** 2)))
    worst_exact = max(worst_exact, rms)
    assert st["not_converged"] == 0 and rms <= 1e-10, (name, rms, st)   # north_star's bar
    pair = []
    for eager in (True, False):
        m = build()
        with smooth.Smoother(m, solver.Option.hip(rtol=1e-30, max_inner=5, check_every=5, eager_scalars=eager), control) as sm:
            sm.iterate(1)
            sm.download()
        pair.append(mesh_flat(m))
    d = float(np.abs(pair[0] - pair[1]).max())
    worst_pair = max(worst_pair, d)
    assert np.isfinite(pair[1]).all() and d <= 1e-11, (name, d)
    print(f"case {k}: {name}: rms vs exact {rms:.2e}, two recurrences differ by {d:.1e}", flush=True)
print(f"{cases} cases: worst rms vs the exact-solve oracle {worst_exact:.2e}, worst difference between the recurrences after 5 updates {worst_pair:.1e}")
