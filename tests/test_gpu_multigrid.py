"""TM_INNER_MG_BICGSTAB: Picard + BiCGStab right-preconditioned by a block-local multigrid V-cycle.
A preconditioner changes the route, not the destination: every Picard iterate must still equal the exact-solve oracle's
(reference semantics with its UMFPACK backend) within 1e-10 RMS, on every topology, and with far fewer inner iterations
than the diagonal-only solver on anything but tiny blocks."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from tests.meshes import TOPOLOGIES
from tests.test_o4h import load
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

pytestmark = pytest.mark.gpu
MG = dict(inner=solver.Inner.mg_bicgstab, rtol=1e-13, max_inner=2000, check_every=1)


def _rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_mg_picard_iterates_match_exact_oracle(name):
    mesh = TOPOLOGIES[name]()
    om = OracleMesh(mesh)
    hist, iterates = oracle.picard_exact(om, 3, keep_iterates=True)
    with smooth.Smoother(mesh, solver.Option.hip(**MG)) as sm:
        for k in range(3):
            st = sm.iterate(1)
            assert st["not_converged"] == 0, st
            sm.download()
            ref = np.concatenate([b.reshape(-1, 2) for b in iterates[k]])
            assert _rms(mesh_flat(mesh), ref) <= 1e-10, (name, k, _rms(mesh_flat(mesh), ref))
            assert st["last_residual"] == pytest.approx(hist[k], rel=1e-6, abs=1e-30)


@pytest.mark.parametrize("ni,nj", [(129, 129), (130, 200), (64, 257), (5, 300), (200, 4), (70001, 9)])
def test_mg_block_shapes_and_iteration_counts(ni, nj):
    # odd / even sizes (an even size coarsens with one short last cell), strong anisotropy in the index space, a direction too
    # short to coarsen: same iterate as the diagonal solver, never more inner iterations
    build = lambda: configs.single_block(ni, nj, perturb=0.25)
    a, b = build(), build()
    with smooth.Smoother(a, solver.Option.hip(**MG)) as sm:
        st_mg = sm.iterate(2)
        sm.download()
    with smooth.Smoother(b, solver.Option.hip(rtol=1e-13, max_inner=20000, check_every=1)) as sm:
        st_d = sm.iterate(2)
        sm.download()
    assert st_mg["not_converged"] == 0
    assert st_mg["inner_iterations"] <= st_d["inner_iterations"]
    if st_d["not_converged"]:   # (70001, 9): cells 1000x longer than wide -- the diagonal alone gives up at 20000 iterations
        assert ni > 65535 and np.all(np.isfinite(mesh_flat(a)))
        return
    assert _rms(mesh_flat(a), mesh_flat(b)) <= 2e-10
    if min(ni, nj) >= 64:
        assert st_mg["inner_iterations"] <= 30, st_mg            # mesh-independent: a handful per Picard solve
        assert st_d["inner_iterations"] >= 5 * st_mg["inner_iterations"]


def test_mg_white_control_function():
    mesh = TOPOLOGIES["plate_le"]()
    om = OracleMesh(mesh)
    oracle.picard_exact(om, 4, control=("white", 0.02, 0.5 * np.pi))
    with smooth.Smoother(mesh, solver.Option.hip(**MG), wcf.Algorithm(wcf.White(0.02))) as sm:
        st = sm.iterate(4)
        sm.download()
    assert st["not_converged"] == 0
    assert _rms(mesh_flat(mesh), om.flat()) <= 1e-10   # north_star's bar; (P,Q) pass through ocml's acos / atan2 (ulp-level)


def test_mg_t106_o4h():
    inp, mesh = load("T106", None)
    om = OracleMesh(mesh)
    oracle.picard_exact(om, 2)
    with smooth.Smoother(mesh, solver.Option.hip(**MG)) as sm:
        st = sm.iterate(2)
        sm.download()
    assert st["not_converged"] == 0
    assert _rms(mesh_flat(mesh), om.flat()) <= 1e-10


def test_iterate_until_residual_tolerance():
    # residual-based stop of the outer iteration (the reference only has a fixed count): the returned mesh satisfies the bound,
    # a second call returns at once without touching it, and the relax mode honours the same bound in steps of 32 sweeps
    mesh = configs.single_block(257, 257, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-10)) as sm:
        reached, st = sm.iterate_until(1e-9, 50)
        assert reached and 1 <= st["outer_iterations"] <= 12 and st["scaled_residual_rms"] <= 1e-9
        sm.download()
        before = mesh_flat(mesh).copy()
        reached, st2 = sm.iterate_until(1e-9, 50)
        sm.download()
        assert reached and st2["outer_iterations"] == 0 and np.array_equal(mesh_flat(mesh), before)
        reached, st3 = sm.iterate_until(1e-30, 2)     # unreachable: stops at the cap and says so
        assert not reached and st3["outer_iterations"] == 2
    small = configs.single_block(33, 33, perturb=0.25)
    with smooth.Smoother(small, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        reached, st = sm.iterate_until(1e-6, 100000)
        assert reached and st["outer_iterations"] % 32 == 0 and st["scaled_residual_rms"] <= 1e-6


@pytest.mark.parametrize("name", ["single_perturbed_33", "strip3_reversed", "channel_periodic_fixed"])
def test_converged_coordinates_match_cpu(name):
    # BASELINE target: CONVERGED coordinates within 1e-10 RMS of the CPU path.  CPU: the oracle's exact-solve Picard iteration
    # run to its fixed point; GPU: iterate_until on the scaled nonlinear residual with the multigrid-preconditioned solver.
    mesh = TOPOLOGIES[name]()
    om = OracleMesh(mesh)
    hist, _ = oracle.picard_exact(om, 40)
    assert hist[-1] <= 1e-40 or hist[-1] <= 1e-24 * max(hist[0], 1e-300)   # (sum dx^2 + sum dy^2)^2 of the last CPU iteration: converged
    with smooth.Smoother(mesh, solver.Option.hip(**MG)) as sm:
        reached, st = sm.iterate_until(1e-13, 60)
        sm.download()
    assert reached, st
    assert _rms(mesh_flat(mesh), om.flat()) <= 1e-10


@pytest.mark.parametrize("shape", [(129, 257), (100, 131), (67, 300), (33, 33), (5, 70), (300, 9)])
def test_prolongation_folded_into_the_post_smoothing_sweep_is_bit_identical(shape, monkeypatch):
    # the correction interpolated as the rows enter K2's window (even / odd rows and columns, wave seams at columns 64 k, semi-coarsened
    # levels of stretched blocks, halo columns) against the stand-alone k_mg_prolong_add: same expression, same bits
    out = []
    monkeypatch.setenv("TM_MG_PAIR", "0")   # the one-sweep-per-pass cycle (the two-per-pass kernels have their own test below)
    for fused in ("2", "0"):   # 2 = the folded form on every level, 0 = never
        monkeypatch.setenv("TM_MG_FUSE_PROLONG", fused)
        mesh = configs.single_block(shape[0], shape[1], perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-12, max_inner=200, check_every=1)) as sm:
            st = sm.iterate(2)
            sm.download()
        assert st["not_converged"] == 0
        out.append((mesh.blocks[0].points.data.copy(), st["inner_iterations"]))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])


def test_folded_prolongation_with_a_control_function_and_many_blocks(monkeypatch):
    # the (P, Q) flavour of the folded kernel and multi-block meshes: the White plate and T106 (8 blocks, semi-coarsened levels)
    for build, control in ((TOPOLOGIES["plate_le"], wcf.Algorithm(wcf.White(0.02))), (lambda: load("T106", None)[1], None)):
        out = []
        monkeypatch.setenv("TM_MG_PAIR", "0")
        for fused in ("2", "0"):
            monkeypatch.setenv("TM_MG_FUSE_PROLONG", fused)
            mesh = build()
            with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-12, max_inner=400, check_every=1), control) as sm:
                st = sm.iterate(2)
                sm.download()
            assert st["not_converged"] == 0
            out.append((mesh_flat(mesh), st["inner_iterations"]))
        assert out[0][1] == out[1][1]
        assert np.array_equal(out[0][0], out[1][0])


PAIR_SHAPES = [(129, 257), (100, 131), (67, 300), (33, 33), (5, 70), (300, 9), (61, 62), (64, 121), (6, 5), (250, 241), (1000, 700), (2049, 1030)]


@pytest.mark.parametrize("shape", PAIR_SHAPES, ids=[f"{a}x{b}" for a, b in PAIR_SHAPES])
def test_two_operator_applications_per_pass_are_bit_identical(shape, monkeypatch):
    # k_mg_pair -- both pre-sweeps + the residual in one pass, the prolongation + both post-sweeps in one pass, on every level of at
    # least 5 x 5 nodes -- against the one-sweep-per-pass kernels (TM_MG_PAIR=0): the same operation sequence per node, so the
    # same bits after whole Picard solves.  Shapes: strip seams at columns 60 k, chunk seams, even / odd sizes (short last
    # coarse cell), a direction too short to coarsen, semi-coarsened levels, levels that fall back to the old kernels (3 x n).
    out = []
    for pair in ("1", "0"):
        monkeypatch.setenv("TM_MG_PAIR", pair)
        mesh = configs.single_block(shape[0], shape[1], perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-12, max_inner=200, check_every=1)) as sm:
            st = sm.iterate(2)
            sm.download()
        assert st["not_converged"] == 0
        out.append((mesh.blocks[0].points.data.copy(), st["inner_iterations"]))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0]), float(np.abs(out[0][0] - out[1][0]).max())


RESTRICT_SHAPES = [(2049, 2049), (2050, 2307), (4100, 1200), (2305, 2052)]


@pytest.mark.parametrize("shape", RESTRICT_SHAPES, ids=[f"{a}x{b}" for a, b in RESTRICT_SHAPES])
def test_restriction_folded_into_the_pre_pass_is_bit_identical(shape, monkeypatch):
    # levels that fill the device: the full weighting rides behind the PRE pass's residual (58-column strips, the residual never
    # stored) -- k_mg_restrict's nine products in k_mg_restrict's order, so the same bits as the two kernels (TM_MG_RESTRICT_FUSED=0).
    # Shapes: odd / even sizes in either direction (the short last coarse cell), strip seams at columns 58 k, chunk seams at rows 18 k.
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("TM_MG_RESTRICT_FUSED", fused)
        mesh = configs.single_block(shape[0], shape[1], perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-12, max_inner=200, check_every=1)) as sm:
            st = sm.iterate(1)
            sm.download()
        assert st["not_converged"] == 0
        out.append((mesh.blocks[0].points.data.copy(), st["inner_iterations"]))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0]), float(np.abs(out[0][0] - out[1][0]).max())


def test_folded_restriction_with_a_control_function(monkeypatch):
    # the (P, Q) flavour: a White plate of two 2049 x 2100 blocks
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("TM_MG_RESTRICT_FUSED", fused)
        mesh = configs.plate(2049, 2100)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-10, max_inner=60, check_every=1), wcf.Algorithm(wcf.White(0.02))) as sm:
            st = sm.iterate(1)
            sm.download()
        out.append((mesh_flat(mesh), st["inner_iterations"]))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])


def test_two_per_pass_with_a_control_function_and_many_blocks(monkeypatch):
    # the (P, Q) flavours (White plate) and multi-block meshes with stretched cells (T106: 8 blocks, semi-coarsened levels)
    for build, control in ((TOPOLOGIES["plate_le"], wcf.Algorithm(wcf.White(0.02))), (lambda: load("T106", None)[1], None)):
        out = []
        for pair in ("1", "0"):
            monkeypatch.setenv("TM_MG_PAIR", pair)
            mesh = build()
            with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-12, max_inner=400, check_every=1), control) as sm:
                st = sm.iterate(2)
                sm.download()
            assert st["not_converged"] == 0
            out.append((mesh_flat(mesh), st["inner_iterations"]))
        assert out[0][1] == out[1][1]
        assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["strip3_reversed", "two_by_two_junction", "channel_periodic_sliding", "strip2_40x300"])
def test_perimeter_step_of_the_preconditioner_changes_the_route_only(name, monkeypatch):
    # round 4 (late): across interfaces the block-local cycle used to leave the perimeter unknowns with their diagonal; now the perimeter rows are
    # applied to the interior corrections behind the cycles (e_p = f_p - (D^-1 A)_pI e_I, Smoother::precondition).  A preconditioner changes
    # the route, not the destination: same Picard iterates as without the step and as the exact-solve oracle, in fewer inner iterations.
    out = {}
    for step in ("0", "1"):
        monkeypatch.setenv("TM_MG_PERIMETER_STEP", step)
        mesh = TOPOLOGIES[name]()
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab)) as sm:
            st = sm.iterate(3)
            sm.download()
        assert st["not_converged"] == 0
        out[step] = (mesh_flat(mesh), st["inner_iterations"])
    om = OracleMesh(TOPOLOGIES[name]())
    oracle.picard_exact(om, 3)
    rms = _rms(out["0"][0], out["1"][0])
    print(f"[perimeter step] {name}: inner iterations {out['0'][1]} -> {out['1'][1]}, iterates {rms:.1e} apart, {_rms(out['1'][0], om.flat()):.1e} from the exact ones")
    assert rms <= 1e-10 and _rms(out["1"][0], om.flat()) <= 1e-10
    assert out["1"][1] <= out["0"][1]


@pytest.mark.gpu
def test_perimeter_step_cuts_the_iterations_on_a_coupled_strip(monkeypatch):
    counts = {}
    for step in ("0", "1"):
        monkeypatch.setenv("TM_MG_PERIMETER_STEP", step)
        mesh = configs.strip(4, 96, 128)
        rng = np.random.default_rng(5)
        for b in mesh.blocks:
            d = b.points.data
            d[1:-1, 1:-1] += 0.25 / d.shape[0] * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab)) as sm:
            st = sm.iterate(2)
        assert st["not_converged"] == 0
        counts[step] = st["inner_iterations"]
    print(f"[perimeter step] strip 4 x 96 x 128: inner iterations {counts['0']} -> {counts['1']}")
    assert counts["1"] <= 0.85 * counts["0"], counts
