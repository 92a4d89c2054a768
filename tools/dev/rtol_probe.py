#!/usr/bin/env python3
"""How close to the exact-solve oracle do tighter inner tolerances get on a stretched strip (10 x 35 x 135: the worst case of
fuzz_picard_topologies.py)?  usage: rtol_probe.py [nb ni nj]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat, oracle_tfi
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
nb, ni, nj = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 35, 135)
om = OracleMesh(configs.strip(nb, ni, nj, tfi=oracle_tfi))
oracle.picard_exact(om, 2)
for rtol in (1e-12, 1e-13, 1e-14, 1e-15, 1e-16):
    for eager in (False, True):
        m = configs.strip(nb, ni, nj)
        with smooth.Smoother(m, solver.Option.hip(rtol=rtol, max_inner=40000, eager_scalars=eager)) as sm:
            st = sm.iterate(2)
            sm.download()
        rms = float(np.sqrt(np.mean((mesh_flat(m) - om.flat()) ** 2)))
        print(f"rtol {rtol:g} {'classic   ' if eager else 'two-kernel'}: rms vs exact {rms:.2e}, inner {st['inner_iterations']}, not converged {st['not_converged']}", flush=True)
