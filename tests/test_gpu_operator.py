"""K2/K4/K5 parity: the matrix-free operator on the MI355X (through the C-ABI handle) vs the
oracle's assembled CSR (reference smooth.zig:421-1165) -- row by row, bit-exact.

The GPU kernels recompute StencilData (smooth.zig:171-216) on the fly and add the 9 terms of a
row in the CSR column order of the reference, with FMA contraction off, so A*v must equal the
CPU mat-vec (BiCGStab.zig:424-435) in every bit, for every row kind."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh
from tests.meshes import TOPOLOGIES
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

pytestmark = pytest.mark.gpu


def _oracle_apply(om, vec, control=None):
    s = oracle.System(om, control)
    s.fill(0)
    s.fill_x_specific()
    ax = s.matvec(vec[:, 0].copy())
    bx = s.rhs_x.copy()
    dx = s.csr().diagonal()
    s.fill_y_specific()
    ay = s.matvec(vec[:, 1].copy())
    by = s.rhs_y.copy()
    dy = s.csr().diagonal()
    return np.stack([ax, ay], 1), np.stack([bx, by], 1), np.stack([dx, dy], 1), s


def _mirror(om, vec, mode, pq=None):
    """Interior rows of every block in the device operation order (oracle/orc_mirror.cpp); NaN on perimeter rows."""
    out = np.full_like(vec, np.nan)
    off = 0
    for blk in om.blocks:
        ni, nj = blk.shape[:2]
        n = ni * nj
        o = np.full((ni, nj, 2), np.nan)
        oracle.mirror_apply_block(mode, vec[off:off + n].reshape(ni, nj, 2), blk, None if pq is None else pq[off:off + n].reshape(ni, nj, 2), out=o)
        out[off:off + n] = o.reshape(n, 2)
        off += n
    return out


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_operator_rows(name):
    """Perimeter rows: bit-exact vs the reference-order CSR.  Interior rows: bit-exact vs the oracle's mirror of the
    device operation order, and within rounding of the CSR mat-vec (tolerance: 16 eps x sum_k |c_k w_k|)."""
    mesh = TOPOLOGIES[name]()
    om = OracleMesh(mesh)
    rng = np.random.default_rng(7)
    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        vec = rng.standard_normal((sm.dof, 2))
        ref, rhs_ref, diag, s = _oracle_apply(om, vec)
        kinds = sm.row_kinds()
        perim = kinds >= 0
        assert np.array_equal(kinds[perim], s.boundary_kind)   # smooth.zig:1168-1174, perimeter order
        got = sm.apply(vec, scaled=False)
        bad = np.flatnonzero((got[perim] != ref[perim]).any(axis=1))
        assert bad.size == 0, f"{bad.size} perimeter rows differ, kinds {kinds[perim][bad[:5]]}"
        mir = _mirror(om, vec, oracle.MIRROR_RAW)
        assert np.array_equal(got[~perim], mir[~perim])
        A = abs(s.csr())
        bound = 16 * np.finfo(float).eps * np.stack([A @ np.abs(vec[:, 0]), A @ np.abs(vec[:, 1])], 1)
        assert np.all(np.abs(got - ref) <= bound + 1e-300)
        # right-hand side (static rows + periodic interface rows, smooth.zig:780-921, 1060-1061)
        assert np.array_equal(sm.rhs(), rhs_ref)
        # row-equilibrated operator D^-1 A: diag == 0 -> 1 (BiCGStab.zig:169-173)
        dinv = np.where(diag == 0.0, 1.0, 1.0 / diag)
        gs = sm.apply(vec, scaled=True)
        assert np.array_equal(gs[perim], (ref * dinv)[perim])
        assert np.array_equal(gs[~perim], _mirror(om, vec, oracle.MIRROR_SCALED)[~perim])
        assert np.all(np.abs(gs - ref * dinv) <= (bound + 1e-300) * np.abs(dinv) * 2)
        # acting on the coordinates themselves (field == vector): the first nonlinear residual
        x = om.flat()
        ref_x, _, _, _ = _oracle_apply(OracleMesh(mesh), x)
        gx = sm.apply(x, scaled=False)
        assert np.array_equal(gx[perim], ref_x[perim]) and np.array_equal(gx[~perim], _mirror(om, x, oracle.MIRROR_RAW)[~perim])


def test_operator_with_white_control_function():
    # non-zero (P,Q): interior rows use (P,Q), non-periodic interface rows (Q,P) (smooth.zig:1082-1083)
    mesh = TOPOLOGIES["plate_le"]()
    om = OracleMesh(mesh)
    control = ("white", 0.01, 0.5 * np.pi)
    rng = np.random.default_rng(3)
    with smooth.Smoother(mesh, solver.Option.hip(), wcf.Algorithm(wcf.White(0.01))) as sm:
        vec = rng.standard_normal((sm.dof, 2))
        ref, rhs_ref, _, s = _oracle_apply(om, vec, control)
        pq = sm.control_function()
        np.testing.assert_allclose(pq, s.control_function, rtol=1e-13, atol=1e-13)   # eq. 6.10, wall_control_function.zig:70-280
        assert np.isfinite(pq).all() and np.isfinite(s.control_function).all() and np.abs(pq).max() > 0
        got = sm.apply(vec)
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-12 * scale   # (P,Q) pass through device sqrt/acos/atan2: tolerance, not bits
        # with the DEVICE's own (P,Q) the interior rows are bit-exact against the mirror
        kinds = sm.row_kinds()
        assert np.array_equal(got[kinds < 0], _mirror(om, vec, oracle.MIRROR_RAW, pq)[kinds < 0])


def test_connection_data_check_and_topology_errors():
    from turbomesh_amd import TmError

    mesh = TOPOLOGIES["strip3_9x12"]()
    mesh.blocks[1].points.data[0, 3, 0] += 1e-9   # break the interface: smooth.zig:265-267 panics, we return Mismatch
    with pytest.raises(TmError) as ei:
        smooth.Smoother(mesh, solver.Option.hip())
    assert ei.value.code == -3
    mesh = TOPOLOGIES["single_17x21"]()
    with pytest.raises(TmError) as ei:   # reference backends are not served: ExternalSolverNotEnabled
        smooth.mesh(mesh, 1, solver.Option(tag=solver.Tag.umfpack))
    assert ei.value.code == -5
