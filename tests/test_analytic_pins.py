"""Reference-INDEPENDENT pins for the coupling rows ("what's missing" 3).

The oracle is a restatement of smooth.zig by the same reader who wrote the device plan, and the reference holds no fixture for a smoothed
mesh: a shared misreading of the interface rows (smooth.zig:994-1105), the junction row (:813-836, 1340-1514) or the periodic right-hand
side (:904-915) would pass every oracle-vs-HIP comparison.  The tests below do not compare two restatements; they assert properties that
follow from the MATHEMATICS of the discrete Winslow system and hold for any correct implementation of the coupling:

 (i)  partition invariance -- the same grid as ONE block or cut into 2 / 3 / 4 coupled blocks (row interfaces, column interfaces, reversed
      ranges, a block stored upside down) has the same Laplace fixed point: an interface row is the interior stencil written across
      two blocks, a slaved row a copy, the junction row the mean of its four diagonal neighbours -- which an interior node of a converged
      merged block also satisfies?  No: the junction row is NOT the interior stencil (smooth.zig:813-836 averages the four diagonal
      neighbours), so the 2 x 2 cut is compared with the merged block only in the affine case (ii), where both hold exactly;
 (ii) an affine grid is a fixed point of every consistent discretisation: cut into the 2 x 2 junction mesh, and wrapped into a periodic
      channel whose shift is the affine image of one period, the first Picard iteration must leave it where it is (residual ~ 0).

Each property is checked on the CPU oracle (exact Picard solves) and on the HIP path."""
import numpy as np
import pytest

from oracle import oracle
from tests import meshes
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import configs
from turbomesh_amd.boundary import Connection, Range, Side
from turbomesh_amd.discrete import Mesh

PARTITIONS = [
    # name, isplits, jsplits, flip_i, flip_j, swap          (grid 25 x 19 on the CPU, scaled for the GPU)
    ("two_rows", (12,), (), (), (), ()),
    ("three_rows_reversed_ranges", (8, 16), (), (), (1,), ()),
    ("four_rows", (6, 12, 18), (), (), (3,), ()),
    ("two_columns", (), (9,), (), (), ()),
    ("three_columns_reversed", (), (6, 12), (1,), (), ()),
    ("three_rows_block_upside_down", (8, 16), (), (1,), (), ()),
]


def _grid(ni, nj, tfi):
    return configs.single_block(ni, nj, tfi=tfi, perturb=0.2).blocks[0].points.data.copy()


def _scaled(splits, n_small, n):
    return tuple(int(round(s * (n - 1) / (n_small - 1))) for s in splits)


def _affine(ni, nj):
    i, j = np.meshgrid(np.arange(ni) / (ni - 1), np.arange(nj) / (nj - 1), indexing="ij")
    return np.stack([1.0 + 2.0 * i + 0.3 * j, -0.5 + 0.4 * i + 1.5 * j], axis=2)


def _periodic_affine(ni, nj):
    """One block, the columns j = 0 and j = nj-1 joined by a periodic self connection whose shift is the affine image of one period."""
    g = _affine(ni, nj)
    m = Mesh()
    m.addBlock("affine_channel", configs.block_from_array(g.copy()))
    shift = g[0, -1] - g[0, 0]
    assert np.abs((g[:, -1] - g[:, 0]) - shift).max() < 1e-15
    m.connections.append(Connection((Range(0, Side.i_min, 0, ni - 1), Range(0, Side.i_max, 0, ni - 1)), (float(shift[0]), float(shift[1]))))
    return m, g


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def _oracle_fixed_point(mesh, iterations=60):
    om = OracleMesh(mesh)
    oracle.smooth_mesh(om, iterations, solver=oracle.SOLVER_DIRECT)
    before = om.flat().copy()
    oracle.smooth_mesh(om, 1, solver=oracle.SOLVER_DIRECT)
    assert np.abs(om.flat() - before).max() < 1e-12, "not converged: raise `iterations`"
    return om


@pytest.mark.parametrize("name,isplits,jsplits,flip_i,flip_j,swap", PARTITIONS, ids=[p[0] for p in PARTITIONS])
def test_partition_invariance_oracle(name, isplits, jsplits, flip_i, flip_j, swap):
    NI, NJ = 25, 19
    g = _grid(NI, NJ, oracle_tfi)
    merged = Mesh()
    merged.addBlock("merged", configs.block_from_array(g.copy()))
    want = _oracle_fixed_point(merged).blocks[0]
    parts = meshes.cut(g, isplits, jsplits, flip_i, flip_j, swap)
    om = _oracle_fixed_point(parts)
    got = meshes.uncut(om, NI, NJ, isplits, jsplits, flip_i, flip_j)
    assert not np.isnan(got).any()
    rms = float(np.sqrt(np.mean((got - want) ** 2)))
    assert rms <= 1e-10, f"{name}: fixed point of the coupled blocks differs from the merged block's: rms {rms:.2e}"
    # the copies of an interface node agree to the last bit or two (a slaved row is a plain copy, smooth.zig:1071-1084)
    assert float(np.sqrt(np.mean((want - g) ** 2))) > 1e-4   # ... and the test is not vacuous: the field did move


def test_affine_grid_is_a_fixed_point_on_the_junction_mesh_oracle():
    NI, NJ = 13, 11
    g = _affine(NI, NJ)
    m = meshes.cut(g, (6,), (5,))
    om = OracleMesh(m)
    st = oracle.smooth_mesh(om, 1, solver=oracle.SOLVER_DIRECT)
    assert st.last_residual < 1e-48, st.last_residual          # (sum dx^2 + sum dy^2)^2, smooth.zig:136
    assert np.abs(meshes.uncut(om, NI, NJ, (6,), (5,)) - g).max() < 1e-13
    # pulled away from it (interior nodes, interface nodes and the junction node alike) it comes back
    rng = np.random.default_rng(3)
    for b in om.blocks:
        b[1:-1, 1:-1] += 0.01 * (rng.random(b[1:-1, 1:-1].shape) - 0.5)
    oracle.smooth_mesh(om, 40, solver=oracle.SOLVER_DIRECT)
    assert np.abs(meshes.uncut(om, NI, NJ, (6,), (5,)) - g).max() < 1e-11


def test_affine_grid_is_a_fixed_point_of_the_periodic_channel_oracle():
    m, g = _periodic_affine(12, 9)
    om = OracleMesh(m)
    st = oracle.smooth_mesh(om, 1, solver=oracle.SOLVER_DIRECT)
    assert st.last_residual < 1e-48, st.last_residual
    assert np.abs(om.blocks[0] - g).max() < 1e-13
    # the periodic pair, displaced TOGETHER by a shift-compatible perturbation, and the interior: back to the affine grid
    rng = np.random.default_rng(4)
    om.blocks[0][1:-1, 1:-1] += 0.01 * (rng.random((10, 7, 2)) - 0.5)
    d = 0.01 * (rng.random((10, 2)) - 0.5)
    om.blocks[0][1:-1, 0] += d
    om.blocks[0][1:-1, -1] += d
    oracle.smooth_mesh(om, 60, solver=oracle.SOLVER_DIRECT)
    assert np.abs(om.blocks[0] - g).max() < 1e-11


def test_a_connection_must_name_the_lower_block_first():
    # smooth.zig:627-631 asserts ranges[0].block <= ranges[1].block (the solved side is the lower block): trading the two ranges of an
    # interface is refused by the oracle and by the device plan alike, not silently reinterpreted
    from turbomesh_amd import _capi

    m = meshes.cut(_affine(9, 8), (4,), (), swap=(0,))
    with pytest.raises(oracle.OracleError):
        oracle.smooth_mesh(OracleMesh(m), 1, solver=oracle.SOLVER_DIRECT)
    import ctypes as C

    md = _capi.MeshDesc(m, with_coordinates=False)
    rows = _capi.tm_plan_rows()
    assert _capi.lib().tm_plan_build(md.ref(), C.byref(rows)) == _capi.TM_E_TOPOLOGY


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
def _hip_fixed_point(mesh, tol=2e-13, cap=400):
    from turbomesh_amd.smoothing import smooth, solver

    with smooth.Smoother(mesh, solver.Option.hip()) as sm:
        reached, st = sm.iterate_until_update(tol, cap)
        sm.download()
    assert reached, st
    return mesh


@pytest.mark.gpu
@pytest.mark.parametrize("name,isplits,jsplits,flip_i,flip_j,swap", PARTITIONS, ids=[p[0] for p in PARTITIONS])
def test_partition_invariance_hip(name, isplits, jsplits, flip_i, flip_j, swap):
    NI, NJ = 73, 55
    g = _grid(NI, NJ, None)
    isp, jsp = _scaled(isplits, 25, NI), _scaled(jsplits, 19, NJ)
    merged = Mesh()
    merged.addBlock("merged", configs.block_from_array(g.copy()))
    want = _hip_fixed_point(merged).blocks[0].points.data
    parts = _hip_fixed_point(meshes.cut(g, isp, jsp, flip_i, flip_j, swap))
    got = meshes.uncut(parts, NI, NJ, isp, jsp, flip_i, flip_j)
    rms = float(np.sqrt(np.mean((got - want) ** 2)))
    assert rms <= 1e-10, f"{name}: rms {rms:.2e}"
    assert float(np.sqrt(np.mean((want - g) ** 2))) > 1e-4


@pytest.mark.gpu
def test_affine_fixed_points_hip():
    from turbomesh_amd.smoothing import smooth, solver

    NI, NJ = 41, 33
    g = _affine(NI, NJ)
    m = meshes.cut(g, (20,), (16,))
    st = smooth.mesh(m, 1, solver.Option.hip())
    assert st["last_residual"] < 1e-44 and np.abs(meshes.uncut(m, NI, NJ, (20,), (16,)) - g).max() < 1e-12
    m, g = _periodic_affine(40, 31)
    st = smooth.mesh(m, 1, solver.Option.hip())
    assert st["last_residual"] < 1e-44 and np.abs(m.blocks[0].points.data - g).max() < 1e-12
    # relaxation sweeps (the bench's kernels: pairs / triples through the perimeter-row kernel) keep them fixed as well
    for mesh, ref, back in ((meshes.cut(_affine(NI, NJ), (20,), (16,)), _affine(NI, NJ), lambda q: meshes.uncut(q, NI, NJ, (20,), (16,))),
                            (_periodic_affine(40, 31)[0], _affine(40, 31), lambda q: q.blocks[0].points.data)):
        smooth.mesh(mesh, 7, solver.Option.hip(inner=solver.Inner.relax))
        assert np.abs(back(mesh) - ref).max() < 1e-13
