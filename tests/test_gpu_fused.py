"""K2x2 (two Jacobi elliptic sweeps per kernel pass) against the one-sweep-per-pass path: same arithmetic in the
same order, so the coordinates must agree BIT FOR BIT for every topology, block shape and sweep count (odd counts
finish with one single sweep).  The single-sweep path itself is pinned bit-exactly to the oracle's mirror in
tests/test_gpu_smooth.py / test_gpu_fullsize.py."""
import numpy as np
import pytest

from tests.conftest import mesh_flat
from tests.meshes import TOPOLOGIES
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu


def _run(builder, sweeps, single, omega=1.0, chunks=None):
    mesh = builder()
    opt = solver.Option.hip(inner=solver.Inner.relax, single_sweep=single, omega=omega)
    with smooth.Smoother(mesh, opt) as sm:
        st = None
        for n in (chunks or [sweeps]):
            st = sm.iterate(n)
        sm.download()
    return mesh_flat(mesh), st


@pytest.mark.parametrize("name", list(TOPOLOGIES))
@pytest.mark.parametrize("sweeps", [2, 5])
def test_fused_pairs_equal_single_sweeps_on_every_topology(name, sweeps):
    a, sa = _run(TOPOLOGIES[name], sweeps, single=True)
    b, sb = _run(TOPOLOGIES[name], sweeps, single=False)
    assert np.all(np.isfinite(a))
    assert np.array_equal(a, b), (name, sweeps, float(np.abs(a - b).max()))
    assert sb["operator_sweeps"] == sweeps == sa["operator_sweeps"]
    # the displacement sums are reduced in a different (still deterministic) order
    assert sb["last_dx2"] == pytest.approx(sa["last_dx2"], rel=1e-11, abs=1e-300)
    assert sb["last_dy2"] == pytest.approx(sa["last_dy2"], rel=1e-11, abs=1e-300)


# block shapes around the tiling edges: 60-column strips (4 per workgroup = 240), 64-row chunks, the 5 x 5 minimum
SHAPES = [(5, 5), (5, 64), (6, 61), (7, 62), (61, 63), (63, 121), (66, 122), (67, 181), (130, 241), (131, 242), (129, 243), (200, 483), (24, 20011), (20011, 24)]


@pytest.mark.parametrize("ni,nj", SHAPES)
def test_fused_pairs_block_shapes(ni, nj):
    build = lambda: configs.single_block(ni, nj, perturb=0.2)
    a, _ = _run(build, 4, single=True, omega=0.9)
    b, _ = _run(build, 4, single=False, omega=0.9)
    assert np.array_equal(a, b), (ni, nj, float(np.abs(a - b).max()))
    assert not np.array_equal(a, mesh_flat(build()))   # the sweeps moved the mesh


def test_fused_pairs_across_calls_and_small_blocks_fall_back():
    build = lambda: configs.strip(3, 33, 70, reverse_odd=True)
    a, _ = _run(build, 9, single=True)
    b, _ = _run(build, 9, single=False, chunks=[1, 2, 3, 3])   # 1 | pair | pair+1 | pair+1
    assert np.array_equal(a, b)
    # a block below 5 x 5 cannot hold the two-sweep window: the handle silently takes single sweeps
    build = lambda: configs.strip(2, 4, 9)
    a, _ = _run(build, 6, single=True)
    b, _ = _run(build, 6, single=False)
    assert np.array_equal(a, b)


def test_fused_pairs_multiblock_large():
    build = lambda: configs.strip(2, 300, 1000, reverse_odd=True)
    a, _ = _run(build, 6, single=True)
    b, _ = _run(build, 6, single=False)
    assert np.array_equal(a, b)


def test_config5_independent_slices_equal_separate_runs():
    # BASELINE configs[4] (pseudo-3D: independent spanwise slices) at a small size: one handle smooths all
    # slices concurrently (batched launches); each must equal its own single-block run bit for bit -- "replicas only"
    nsl, n = 5, 97
    mesh = configs.slices(nsl, n, perturb=0.2)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        sm.iterate(7)
        sm.download()
    for k in range(nsl):
        one = configs.slices(1, n, first=k, perturb=0.2)
        with smooth.Smoother(one, solver.Option.hip(inner=solver.Inner.relax)) as sm:
            sm.iterate(7)
            sm.download()
        assert np.array_equal(mesh.blocks[k].points.data, one.blocks[0].points.data), k
    # and through the Krylov path (shared scalars: the slices' systems are solved as one block-diagonal system)
    a = configs.slices(3, 65, perturb=0.2)
    with smooth.Smoother(a, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-13)) as sm:
        st = sm.iterate(2)
        sm.download()
    assert st["not_converged"] == 0
    for k in range(3):
        one = configs.slices(1, 65, first=k, perturb=0.2)
        with smooth.Smoother(one, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-13)) as sm:
            sm.iterate(2)
            sm.download()
        assert float(np.sqrt(np.mean((a.blocks[k].points.data - one.blocks[0].points.data) ** 2))) <= 2e-10


SMALL_COUPLED = {
    "strip3_reversed_20x24": lambda: configs.strip(3, 20, 24, reverse_odd=True),
    "strip2_17x70": lambda: configs.strip(2, 17, 70),
    "junction_20x22": lambda: configs.two_by_two(20, 22),
    "channel_periodic_sliding_21x25": lambda: configs.periodic_channel(21, 25),
    "channel_periodic_fixed_24x19": lambda: configs.periodic_channel(24, 19, sliding=False),
    "junction_33x16": lambda: configs.two_by_two(33, 16),
}


@pytest.mark.parametrize("name", list(SMALL_COUPLED))
@pytest.mark.parametrize("sweeps", [3, 10])
def test_default_schedule_on_small_coupled_meshes_is_the_triple_path_and_bit_identical(name, sweeps, monkeypatch):
    # Round 4: with the three level passes in one launch (k_edge_levels3) coupled sweep triples win at every size, so the library's own
    # default -- no thresholds in the environment -- takes them on any coupled mesh whose blocks have at least 16 x 16 nodes.  Here on the
    # irregular topologies at small sizes: junction rows, periodic and sliding rows, reversed interfaces all pass through the strips'
    # host-computed closures; the result must equal single sweeps bit for bit, with the fused level kernel and with three launches.
    monkeypatch.delenv("TM_TRIPLES_SINGLE_MIN_NODES", raising=False)
    monkeypatch.delenv("TM_TRIPLES_MIN_NODES", raising=False)
    ref = SMALL_COUPLED[name]()
    smooth.mesh(ref, sweeps, solver.Option.hip(inner=solver.Inner.relax, single_sweep=True))
    for fused in ("1", "0"):
        monkeypatch.setenv("TM_LEVELS_FUSED", fused)
        got = SMALL_COUPLED[name]()
        with smooth.Smoother(got, solver.Option.hip(inner=solver.Inner.relax)) as sm:
            sm.iterate(sweeps)
            assert sm.queue_ordering()[0] >= 0, "the two-queue triple schedule did not run"
            sm.download()
        assert np.array_equal(mesh_flat(got), mesh_flat(ref)), (name, sweeps, fused)
