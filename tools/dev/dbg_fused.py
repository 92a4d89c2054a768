import sys, numpy as np
sys.path.insert(0, '.')
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
from tests.conftest import mesh_flat
for n in (2048,):
    res = {}
    for single in (True, False):
        mesh = configs.single_block(n, n)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)) as sm:
            st = sm.iterate(2)
            sm.download()
        res[single] = (mesh.blocks[0].points.data.copy(), st)
    a, b = res[True][0], res[False][0]
    from oracle import oracle
    ref = configs.single_block(n, n).blocks[0].points.data.copy(); oracle.time_relax_sweeps(ref, 2, 1.0)
    print('single==oracle', np.array_equal(a, ref), 'fused==oracle', np.array_equal(b, ref))
    bad = np.argwhere(np.any(a != b, axis=2))
    print(n, 'equal', np.array_equal(a, b), 'nbad', len(bad), 'resid', res[True][1]['last_dx2'], res[False][1]['last_dx2'], res[True][1]['last_dy2'], res[False][1]['last_dy2'])
    if len(bad):
        print(' rows', np.unique(bad[:, 0])[:20], ' cols', np.unique(bad[:, 1])[:20], 'maxdiff', np.abs(a - b).max())
