#!/usr/bin/env python3
"""One block shape: both BiCGStab recurrences against the exact-solve oracle.  usage: shape_probe.py ni nj [iterations = 2] [rtol = 1e-12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
ni, nj = int(sys.argv[1]), int(sys.argv[2])
its = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
om = OracleMesh(configs.single_block(ni, nj, perturb=0.2))
oracle.picard_exact(om, its)
for eager in (True, False):
    m = configs.single_block(ni, nj, perturb=0.2)
    with smooth.Smoother(m, solver.Option.hip(rtol=rtol, max_inner=20000, eager_scalars=eager)) as sm:
        st = sm.iterate(its)
        sm.download()
    d = mesh_flat(m) - om.flat()
    print(f"{ni}x{nj} {'classic' if eager else 'two-kernel'}: rms vs exact {np.sqrt(np.mean(d ** 2)):.3e}, max {np.abs(d).max():.3e}, inner {st['inner_iterations']}, not converged {st['not_converged']}")
