"""CPU: the extended-precision residual evaluation of tests/residual_check.py against the oracle's own exact solve (small sizes)."""
import numpy as np

from oracle import oracle
from tests import residual_check as rc
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import configs


def test_exact_iterate_has_a_rounding_level_residual_and_a_perturbed_one_does_not():
    n = 65
    mesh = configs.single_block(n, n, tfi=oracle_tfi, perturb=0.25)
    x0 = mesh.blocks[0].points.data.copy()
    p, i, v, b = rc.assemble([x0])
    assert len(p) == n * n + 1 and p[-1] == len(v) == 9 * (n - 2) ** 2 + (n * n - (n - 2) ** 2)   # nine per interior row, one per wall row
    om = OracleMesh(mesh)
    oracle.smooth_mesh(om, 1, solver=oracle.SOLVER_DIRECT)            # X^1 by the banded LU (the reference's UMFPACK semantics)
    x1 = om.blocks[0].reshape(-1, 2)
    both, per = rc.relative_residual(p, i, v, b, x1)
    assert both < 1e-14 and max(per) < 1e-14, (both, per)              # the exact solve in fp64: a few eps (1.5e-15 here)
    both64, _ = rc.relative_residual(p, i, v, b, x1, dtype=np.float64)
    assert both64 < 2e-14
    # the storage floor: what a vector picks up from being rounded to fp64 -- the exact solve sits within a small factor of it
    _, _, floor = rc.relative_residual(p, i, v, b, x1, with_floor=True)
    assert 1e-17 < floor < 1e-15 and both < 20 * floor, (both, floor)
    # chunking does not change the figure
    n1, d1 = rc.scaled_residual(p, i, v, b, x1, chunk_rows=1000)
    n2, d2 = rc.scaled_residual(p, i, v, b, x1, chunk_rows=1 << 20)
    assert np.allclose(np.asarray(n1, dtype=float), np.asarray(n2, dtype=float), rtol=1e-12) and np.allclose(np.asarray(d1, dtype=float), np.asarray(d2, dtype=float), rtol=1e-15)
    # X^0 itself is far from solving its own frozen system, and a 1e-9 nudge of X^1 shows at the 1e-9 level
    assert rc.relative_residual(p, i, v, b, x0.reshape(-1, 2))[0] > 1e-4
    bumped = x1.copy()
    bumped[(n // 2) * n + n // 2] += 1e-9
    assert 1e-12 < rc.relative_residual(p, i, v, b, bumped)[0] < 1e-8
