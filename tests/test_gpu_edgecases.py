"""Inputs at the edges of the arithmetic: the degenerate cases the reference's solvers special-case and the wave-uniform slow
paths of the device kernels.

  * collapsed cells -> zero metric -> a_ii = 0: the reference scales such a row by 1 instead of 1/a_ii (BiCGStab.zig:169-173,
    `updateDiagonalInverse`); K2 / K2x2 take the full-division branch of `recip_diag` for the whole wave;
  * spacings so small / large that the fast reciprocal refinement must not be used (D outside [2^-700, 2^700]);
  * non-finite coordinates travel through without hanging anything;
  * the smallest legal blocks (3 x 3: one interior node) and the 5 x 5 minimum of the two-sweep pass."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.discrete import Mesh
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu


def _block(arr):
    m = Mesh()
    m.addBlock("b", configs.block_from_array(np.ascontiguousarray(arr)))
    return m


def _relax(arr, sweeps, single, omega=1.0):
    mesh = _block(arr.copy())
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single, omega=omega)) as sm:
        st = sm.iterate(sweeps)
        sm.download()
    return mesh.blocks[0].points.data.copy(), st


def _mirror(arr, sweeps, omega=1.0):
    ref = np.ascontiguousarray(arr).copy()
    oracle.time_relax_sweeps(ref, sweeps, omega)
    return ref


def test_collapsed_cells_zero_diagonal():
    # a patch of coincident nodes: metric and diagonal vanish there (a_ii = 0 -> D^-1 := 1, BiCGStab.zig:169-173)
    base = configs.single_block(70, 131, perturb=0.2).blocks[0].points.data.copy()
    base[20:27, 40:90] = base[23, 60]          # 7 x 50 nodes on one point
    base[50:53, 3:8] = base[51, 5]
    for sweeps in (1, 2, 5):
        ref = _mirror(base, sweeps, 0.9)
        for single in (True, False):
            got, st = _relax(base, sweeps, single, 0.9)
            assert np.isfinite(got).all()
            assert np.array_equal(got, ref), (sweeps, single)
    # the operator itself against the faithfully assembled CSR: rows of the collapsed patch are all-zero rows
    mesh = _block(base.copy())
    om = OracleMesh(mesh)
    s = oracle.System(om)
    s.fill(0)
    A = s.csr()
    rng = np.random.default_rng(5)
    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        vec = rng.standard_normal((sm.dof, 2))
        got = sm.apply(vec)
        gs = sm.apply(vec, scaled=True)
    ref = np.stack([A @ vec[:, 0], A @ vec[:, 1]], 1)
    diag = A.diagonal()
    assert (diag == 0).sum() >= 5 * 48 + 3                      # the interior of the patches
    bound = 16 * np.finfo(float).eps * np.stack([abs(A) @ np.abs(vec[:, 0]), abs(A) @ np.abs(vec[:, 1])], 1)
    assert np.all(np.abs(got - ref) <= bound + 1e-300)
    dinv = np.where(diag == 0.0, 1.0, 1.0 / np.where(diag == 0.0, 1.0, diag))
    assert np.all(np.abs(gs - ref * dinv[:, None]) <= (bound + 1e-300) * np.abs(dinv[:, None]) * 2)
    zero_rows = np.flatnonzero(diag == 0)
    assert np.all(got[zero_rows] == 0) and np.all(gs[zero_rows] == 0)
    s.close()


@pytest.mark.parametrize("scale", [2.0 ** -400, 2.0 ** -352, 2.0 ** 330])
def test_extreme_spacings_take_the_exact_division(scale):
    # squared spacings below 2^-700: recip_diag must not use the v_rcp_f64 refinement (its scalings are not the identity there); the
    # whole wave takes the IEEE division and still matches the mirror bit for bit.  (Above 2^700 the row sums G * w overflow before
    # the reciprocal matters; 2^330 stays finite and exercises the fast path at the top of its range.)
    base = configs.single_block(40, 131, perturb=0.2).blocks[0].points.data * scale
    for single in (True, False):
        got, _ = _relax(base, 4, single)
        assert np.isfinite(got).all() and np.array_equal(got, _mirror(base, 4)), (scale, single)


def test_non_finite_coordinates_do_not_hang():
    base = configs.single_block(33, 70, perturb=0.2).blocks[0].points.data.copy()
    base[10, 20, 0] = np.nan
    base[20, 40, 1] = np.inf
    for single in (True, False):
        got, st = _relax(base, 4, single)
        ref = _mirror(base, 4)
        assert np.array_equal(np.isfinite(got), np.isfinite(ref))
        ok = np.isfinite(ref)
        assert np.array_equal(got[ok], ref[ok])
        assert not np.isfinite(st["last_residual"])
    mesh = _block(base.copy())
    st = smooth.mesh(mesh, 1, solver.Option.hip(max_inner=50))    # a Picard solve on it ends (breakdown / max_inner), it does not spin
    assert st["outer_iterations"] == 1


@pytest.mark.parametrize("ni,nj", [(3, 3), (3, 9), (4, 4), (5, 5), (5, 3)])
def test_smallest_blocks(ni, nj):
    base = configs.single_block(max(ni, 5), max(nj, 5), perturb=0.3).blocks[0].points.data[:ni, :nj].copy()
    ref = _mirror(base, 3)
    for single in (True, False):      # below 5 x 5 the two-sweep pass is not available: the handle takes single sweeps
        got, st = _relax(base, 3, single)
        assert np.array_equal(got, ref) and st["operator_sweeps"] == 3
    om = OracleMesh(_block(base.copy()))
    oracle.picard_exact(om, 2)
    mesh = _block(base.copy())
    st = smooth.mesh(mesh, 2, solver.Option.hip(rtol=1e-13))
    assert st["not_converged"] == 0 and float(np.sqrt(np.mean((mesh_flat(mesh) - om.flat()) ** 2))) <= 1e-10


def test_empty_and_undersized_inputs_are_refused_not_crashed():
    # an empty mesh, a block with no interior node, a connection shorter than the reference's iterators allow (lenInternal() > 3 is
    # asserted there, smooth.zig:627-631): error codes through the C ABI, never an abort or a launch
    from turbomesh_amd import _capi
    from turbomesh_amd.boundary import Connection, Range, Side

    with pytest.raises(_capi.TmError) as e:
        smooth.mesh(Mesh(), 1, solver.Option.hip())
    assert e.value.code in (_capi.TM_E_ARG, _capi.TM_E_SIZE)
    for shape in ((2, 9), (9, 2), (1, 1)):
        arr = np.zeros(shape + (2,))
        with pytest.raises(_capi.TmError) as e:
            smooth.mesh(_block(arr), 1, solver.Option.hip())
        assert e.value.code in (_capi.TM_E_ARG, _capi.TM_E_SIZE, _capi.TM_E_UNSUPPORTED), shape
    # zero iterations: nothing moves, nothing is launched that could fail
    base = configs.single_block(9, 11, perturb=0.2).blocks[0].points.data.copy()
    mesh = _block(base.copy())
    st = smooth.mesh(mesh, 0, solver.Option.hip())
    assert st["outer_iterations"] == 0 and np.array_equal(mesh.blocks[0].points.data, base)
    # two blocks whose connection names ranges of different lengths
    two = configs.strip(2, 9, 12)
    c = two.connections[0]
    two.connections[0] = Connection((c.ranges[0], Range(c.ranges[1].block, c.ranges[1].side, 0, 7)), c.periodicity)
    with pytest.raises(_capi.TmError) as e:
        smooth.mesh(two, 1, solver.Option.hip())
    assert e.value.code in (_capi.TM_E_TOPOLOGY, _capi.TM_E_MISMATCH, _capi.TM_E_SIZE)
