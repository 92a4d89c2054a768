"""The reference's ASSEMBLED system on the device: tm_smoother_assemble_csr fills RowCompressedMatrixSystem2d -- pattern
(smooth.zig:309-385) and values (system.fill, smooth.zig:923-1113; StencilData.init in the reference's expression order, smooth.zig:171-216)
-- from the coordinates resident on the device, and tm_smoother_apply_reference_order multiplies through it row by row in CSR order
(BiCGStab.zig:424-435).

Against the FAITHFUL oracle (oracle/orc_system.cpp, the restatement of those very lines) everything is bit for bit -- pattern, every
coefficient of every row, the product on every row, interior rows included -- on the ten topologies.  This is the reference-order
operator on the device; the matrix-free fast path (tm_smoother_apply) shares its perimeter rows bit for bit and evaluates interior rows in
a factored form within 16 eps sum |c_k w_k| of it."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh
from tests.meshes import TOPOLOGIES
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

pytestmark = pytest.mark.gpu


def _oracle_system(mesh, control=None, iteration=0):
    om = OracleMesh(mesh)
    s = oracle.System(om, control)
    s.fill(iteration)
    s.fill_x_specific()
    vx = s.lhs_values.copy()
    s.fill_y_specific()
    vy = s.lhs_values.copy()
    return om, s, vx, vy


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_device_assembled_system_equals_the_faithful_oracle_bit_for_bit(name):
    mesh = TOPOLOGIES[name]()
    om, s, vx, vy = _oracle_system(mesh)
    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        Ap, Ai, Ax, Ay = sm.assemble_csr()
        assert np.array_equal(Ap, s.lhs_p) and np.array_equal(Ai, s.lhs_i), name            # RowCompressedMatrixSystem2d.init
        assert np.array_equal(Ax, vx), f"{name}: {np.count_nonzero(Ax != vx)} of {len(vx)} x-system values differ"
        assert np.array_equal(Ay, vy), f"{name}: {np.count_nonzero(Ay != vy)} y-system values differ"
        assert np.array_equal(sm.rhs(), np.stack([s.rhs_x, s.rhs_y], axis=1))
        # the product through it: the reference's mat-vec on EVERY row, interior rows included
        w = np.random.default_rng(5).standard_normal((sm.dof, 2))
        got = sm.apply_reference_order(w)
        s.fill_x_specific()
        rx = s.matvec(w[:, 0].copy())
        s.fill_y_specific()
        ry = s.matvec(w[:, 1].copy())
        assert np.array_equal(got[:, 0], rx) and np.array_equal(got[:, 1], ry), name
        # and the fast path against it: identical on the perimeter rows, within 16 eps sum |c w| on the interior rows
        fast = sm.apply(w, scaled=False)
        kinds = sm.row_kinds()
        assert np.array_equal(fast[kinds >= 0], got[kinds >= 0])
        import scipy.sparse as sp

        absA = abs(sp.csr_matrix((Ax, Ai, Ap), shape=(sm.dof, sm.dof)))
        bound = 16 * np.finfo(float).eps * np.stack([absA @ np.abs(w[:, 0]), absA @ np.abs(w[:, 1])], axis=1)
        assert np.all(np.abs(fast - got) <= bound + 1e-300)


def test_assembled_system_follows_the_coordinates_and_the_control_function():
    # after Picard iterations the device's coordinates have moved: the assembly is of the CURRENT field; with the White control function the
    # interior rows carry (P, Q) and non-periodic interface rows (Q, P) (smooth.zig:1082-1083) -- compared with the oracle stepped alongside
    mesh = TOPOLOGIES["plate_le"]()
    control = ("white", 0.02, 0.5 * np.pi)
    om = OracleMesh(mesh)
    s = oracle.System(om, control)
    with smooth.Smoother(mesh, solver.Option.hip(), wcf.Algorithm(wcf.White(0.02))) as sm:
        s.fill(0)
        pq = sm.control_function()
        np.testing.assert_allclose(pq, s.control_function, rtol=1e-13, atol=1e-13)
        Ap, Ai, Ax, Ay = sm.assemble_csr()
        assert np.array_equal(Ap, s.lhs_p) and np.array_equal(Ai, s.lhs_i)
        s.fill_x_specific()
        # (P, Q) pass through sqrt / acos / atan2 on the device: the coefficients agree to rounding of those, not necessarily to the bit
        np.testing.assert_allclose(Ax, s.lhs_values, rtol=1e-12, atol=1e-300)
    plain = TOPOLOGIES["single_perturbed_33"]()
    with smooth.Smoother(plain, solver.Option.hip()) as sm:
        before = sm.assemble_csr()[2].copy()
        sm.iterate(2)
        after = sm.assemble_csr()[2]
        assert not np.array_equal(before, after)
        sm.download()
    # the ORACLE assembled at the device's own iterate (a block without connections: nothing for connectionDataCheck to compare): bits again
    om3, s3, vx3, vy3 = _oracle_system(plain)
    with smooth.Smoother(plain, solver.Option.hip()) as sm:
        Ap, Ai, Ax, Ay = sm.assemble_csr()
    assert np.array_equal(Ax, vx3) and np.array_equal(Ay, vy3) and np.array_equal(Ax, after)


def test_size_query_capacity_and_multi_rank_refusal():
    import ctypes as C

    from turbomesh_amd import _capi

    mesh = configs.single_block(9, 11)
    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        n = C.c_uint64(0)
        assert _capi.lib().tm_smoother_assemble_csr(sm._h, None, None, None, None, 0, C.byref(n)) == 0
        assert n.value == 9 * (7 * 9) + (99 - 63)          # nine per interior row, one per (fixed) perimeter row
        ax = np.empty(5)
        assert _capi.lib().tm_smoother_assemble_csr(sm._h, None, None, _capi.f64ptr(ax), None, 5, C.byref(n)) == _capi.TM_E_SIZE
