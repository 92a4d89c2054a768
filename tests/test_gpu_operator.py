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


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_operator_rows_bit_exact(name):
    mesh = TOPOLOGIES[name]()
    om = OracleMesh(mesh)
    rng = np.random.default_rng(7)
    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        vec = rng.standard_normal((sm.dof, 2))
        ref, rhs_ref, diag, s = _oracle_apply(om, vec)
        got = sm.apply(vec, scaled=False)
        bad = np.flatnonzero((got != ref).any(axis=1))
        assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5]}, kinds {sm.row_kinds()[bad[:5]]}"
        # right-hand side (static rows + periodic interface rows, smooth.zig:780-921, 1060-1061)
        assert np.array_equal(sm.rhs(), rhs_ref)
        # row-equilibrated operator: sum * (1/diag), diag == 0 -> 1 (BiCGStab.zig:169-173)
        dinv = np.where(diag == 0.0, 1.0, 1.0 / diag)
        assert np.array_equal(sm.apply(vec, scaled=True), ref * dinv)
        # row kinds (smooth.zig:1168-1174) in perimeter order
        kinds = sm.row_kinds()
        assert np.array_equal(kinds[kinds >= 0], s.boundary_kind)
        # acting on the coordinates themselves (field == vector): first nonlinear residual
        x = om.flat()
        ref_x, _, _, _ = _oracle_apply(OracleMesh(mesh), x)
        assert np.array_equal(sm.apply(x, scaled=False), ref_x)


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
        assert np.abs(got - ref).max() <= 1e-12 * scale   # P,Q may differ in the last bit (device division order is IEEE, but kept as tolerance)


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
