"""Entry points added in round 3 (include/tm_hip.h): tm_smoother_iterate_until_update, tm_stream_probe; the three-sweeps-per-pass path."""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import mesh_flat
from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu


def test_iterate_until_update_stops_on_the_picard_update_and_reports_the_cap():
    # the reference's own per-iteration quantity (smooth.zig:112-137): sqrt((sum dx^2 + sum dy^2) / nodes) of the last outer iteration
    n = 257
    mesh = configs.single_block(n, n, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab)) as sm:
        reached, st = sm.iterate_until_update(1e-9, 100)
        upd = np.sqrt((st["last_dx2"] + st["last_dy2"]) / (n * n))
        assert reached and upd <= 1e-9 and 2 <= st["outer_iterations"] <= 40, st
        reached2, st2 = sm.iterate_until_update(1e-30, 2)      # unreachable: stops at the cap and says so
        assert not reached2 and st2["outer_iterations"] == 2
    relax = configs.single_block(33, 33, perturb=0.25)
    with smooth.Smoother(relax, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        reached, st = sm.iterate_until_update(1e-9, 100000)   # relax mode tests every 32 sweeps
        assert reached and st["outer_iterations"] % 32 == 0
        assert np.sqrt((st["last_dx2"] + st["last_dy2"]) / (33 * 33)) <= 1e-9
    with pytest.raises(_capi.TmError):
        with smooth.Smoother(configs.single_block(9, 9)) as sm:
            sm.iterate_until_update(0.0, 3)


def test_stream_probe_reports_a_plausible_ceiling():
    cp, tr = C.c_double(0), C.c_double(0)
    _capi.check(_capi.lib().tm_stream_probe(64 << 20, 10, C.byref(cp), C.byref(tr)))
    assert 2000.0 < cp.value < 12000.0 and 2000.0 < tr.value < 12000.0, (cp.value, tr.value)
    assert _capi.lib().tm_stream_probe(16, 10, C.byref(cp), C.byref(tr)) == _capi.TM_E_ARG


@pytest.mark.parametrize("shape", [(7, 7), (8, 61), (64, 59), (70, 131), (131, 300), (40, 1000)])
def test_three_sweeps_per_pass_equal_single_sweeps(shape, monkeypatch):
    # K2x3 (blocks whose perimeter rows are all fixed) against one sweep per pass, every split of the sweep count into triples, pairs
    # and single sweeps, around the 58-column strip seams and the chunk seams; and with omega != 1 (the general update form)
    unit = None
    for omega in (0.0, 0.7):
        out = []
        for single in (True, False):
            mesh = configs.single_block(shape[0], shape[1], perturb=0.2)
            hist = []
            with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single, omega=omega)) as sm:
                for nsweeps in (3, 4, 5, 6, 1, 2):
                    st = sm.iterate(nsweeps)
                    sm.download()
                    hist.append((mesh_flat(mesh).copy(), st["last_dx2"], st["last_dy2"]))
            out.append(hist)
        for (a, ax, ay), (b, bx, by) in zip(*out):
            assert np.array_equal(a, b), float(np.abs(a - b).max())
            assert bx == pytest.approx(ax, rel=1e-10, abs=1e-300) and by == pytest.approx(ay, rel=1e-10, abs=1e-300)
        if omega == 0.0:
            unit = out[0][-1][0]
    # the knob: TM_FUSE_3=0 gives pairs, same bits
    monkeypatch.setenv("TM_FUSE_3", "0")
    mesh = configs.single_block(shape[0], shape[1], perturb=0.2)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        for nsweeps in (3, 4, 5, 6, 1, 2):
            sm.iterate(nsweeps)
        sm.download()
    assert np.array_equal(mesh_flat(mesh), unit)


def test_relax_sweep_with_the_white_control_function_matches_the_mirror():
    # MODE_RELAX with (P, Q) and omega = 1 takes the Jacobi form x_new = q / (2 D) with the control-function terms inside q
    # (winslow_row); the oracle's mirror makes the same step: interior rows of both blocks of the White plate, bit for bit
    from oracle import oracle
    from tests.meshes import TOPOLOGIES
    from turbomesh_amd.smoothing import wall_control_function as wcf

    mesh = TOPOLOGIES["plate_le"]()
    x0 = [b.points.data.copy() for b in mesh.blocks]
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=True), wcf.Algorithm(wcf.White(0.02))) as sm:
        pq = sm.control_function()
        sm.iterate(1)
        sm.download()
    assert np.abs(pq).max() > 0
    off = 0
    for b, x in zip(mesh.blocks, x0):
        ni, nj = x.shape[:2]
        pqb = np.ascontiguousarray(pq[off:off + ni * nj].reshape(ni, nj, 2))
        off += ni * nj
        ref = oracle.mirror_apply_block(oracle.MIRROR_RELAX, x, x, pq=pqb, omega=1.0, out=x.copy())
        assert np.array_equal(b.points.data[1:-1, 1:-1], ref[1:-1, 1:-1]), float(np.abs(b.points.data[1:-1, 1:-1] - ref[1:-1, 1:-1]).max())
        assert not np.array_equal(b.points.data[1:-1, 1:-1], x[1:-1, 1:-1])


def test_inexact_picard_reaches_the_same_fixed_point_with_fewer_inner_iterations():
    # TM_OPT_RTOL_INITIAL: the inner tolerance relative to each solve's INITIAL residual.  Every solve iterates (a tolerance relative
    # to ||D^-1 b|| that the warm start already meets would return at once and fake a zero update), the fixed point is the same,
    # the inner iterations on the way are several times fewer.
    n = 513
    res = []
    for kw in (dict(), dict(rtol=1e-2, rtol_initial=True)):
        mesh = configs.single_block(n, n, perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, check_every=1, **kw)) as sm:
            reached, st = sm.iterate_until_update(1e-11, 200)
            sm.download()
        assert reached and st["not_converged"] == 0, st
        res.append((mesh.blocks[0].points.data.copy(), st["inner_iterations"], st["outer_iterations"]))
        print(f"[inexact picard] {kw or 'default'}: outer {st['outer_iterations']}, inner {st['inner_iterations']}, {st['seconds'] * 1e3:.1f} ms")
    assert float(np.sqrt(np.mean((res[0][0] - res[1][0]) ** 2))) <= 1e-9
    assert res[1][1] * 2 <= res[0][1] and res[1][1] >= res[1][2]          # at least one inner iteration per outer one


@pytest.mark.parametrize("kind", ["strip3", "strip2_reversed", "two_by_two"])
def test_three_sweeps_per_pass_on_coupled_blocks(kind, monkeypatch):
    # Coupled blocks of a single process: K2x3 with a frozen perimeter stores everything but the nodes within two of a side whose
    # perimeter rows move; the perimeter-row kernel evaluates the perimeter and that zone level by level (Smoother::
    # relax_triples_coupled).  Bit-identical to single sweeps: row interfaces, reversed ranges, column interfaces and a junction
    # point (two_by_two), every split of the sweep count into triples, pairs and single sweeps.
    build = {"strip3": lambda: configs.strip(3, 900, 800), "strip2_reversed": lambda: configs.strip(2, 1100, 1000, reverse_odd=True),
             "two_by_two": lambda: configs.two_by_two(760, 720)}[kind]
    hist = {}
    for mode in ("triples", "single"):
        mesh = build()
        rng = np.random.default_rng(3)
        for b in mesh.blocks:   # roughen the interiors (the interfaces stay matched)
            d = b.points.data
            d[1:-1, 1:-1] += 0.2 / d.shape[0] * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
        out = []
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=(mode == "single"))) as sm:
            for nsweeps in (3, 7, 5, 6):
                st = sm.iterate(nsweeps)
                sm.download()
                out.append((mesh_flat(mesh).copy(), st["last_dx2"], st["last_dy2"]))
        hist[mode] = out
    for (a, ax, ay), (b, bx, by) in zip(hist["triples"], hist["single"]):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (kind, float(np.abs(a - b).max()), int(np.any(a != b, axis=1).sum()))
        assert ax == pytest.approx(bx, rel=1e-10, abs=1e-300) and ay == pytest.approx(by, rel=1e-10, abs=1e-300)
    monkeypatch.setenv("TM_TRIPLES_COUPLED", "0")   # the knob: pairs, same bits
    mesh = build()
    rng = np.random.default_rng(3)
    for b in mesh.blocks:
        d = b.points.data
        d[1:-1, 1:-1] += 0.2 / d.shape[0] * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        sm.iterate(3)
        sm.download()
    assert np.array_equal(mesh_flat(mesh), hist["single"][0][0])


def test_inner_auto_is_decided_by_the_largest_block():
    # solver.zig:18-27's option shape with one more payload value: `auto` resolves at create -- the plain solve on meshes of small blocks
    # (the reference's examples), the multigrid-preconditioned one from 100 000 nodes per block on; explicit choices are kept
    from turbomesh_amd import configs
    from turbomesh_amd.smoothing import smooth, solver

    auto = solver.Option.hip(inner=solver.Inner.auto)
    # (round 4, late: a block that no connection couples takes the cycle from 1000 nodes on -- it is block-local and needs ~25 iterations there at
    # any size; coupled blocks keep the 100 000-node rule: tools/dev/auto_crossover.py)
    for mesh, want in ((configs.single_block(221, 41), solver.Inner.mg_bicgstab), (configs.single_block(31, 32), solver.Inner.bicgstab),
                       (configs.strip(3, 40, 50), solver.Inner.bicgstab), (configs.strip(2, 316, 316), solver.Inner.bicgstab),
                       (configs.strip(2, 317, 317), solver.Inner.mg_bicgstab), (configs.single_block(317, 317), solver.Inner.mg_bicgstab),
                       (configs.strip(2, 64, 2000), solver.Inner.mg_bicgstab)):
        with smooth.Smoother(mesh, auto) as sm:
            assert sm.inner == want
    with smooth.Smoother(configs.single_block(400, 400), solver.Option.hip(inner=solver.Inner.bicgstab)) as sm:
        assert sm.inner == solver.Inner.bicgstab
    # and both reach the same Picard iterates (a preconditioner changes the route, not the destination)
    a, b = configs.single_block(330, 330, perturb=0.2), configs.single_block(330, 330, perturb=0.2)
    smooth.mesh(a, 2, auto)
    smooth.mesh(b, 2, solver.Option.hip(inner=solver.Inner.bicgstab))
    import numpy as np

    assert float(np.sqrt(np.mean((a.blocks[0].points.data - b.blocks[0].points.data) ** 2))) <= 1e-10


def test_inner_auto_keeps_the_plain_solve_on_boundary_layer_clustering():
    # round 4: a block-wide point-Jacobi cycle is a poor preconditioner where the cells' aspect ratio varies strongly inside a block (the
    # reference's O-grids refined to 10^5..10^6 nodes: tools/dev/o4h_auto_probe.py); a single-process handle reads that off the coordinates
    import json, os

    from turbomesh_amd import clustering, configs
    from turbomesh_amd.input import Input
    from turbomesh_amd.smoothing import smooth, solver

    auto = solver.Option.hip(inner=solver.Inner.auto)
    n = 400
    # a rectangle with wall clustering in j (first cell 1e-4 of the height): aspect ratios from ~600 at the wall to ~0.2 at the far side
    mesh = configs.single_block(n, n)
    y = clustering.SingleHyperbolicClustering(1e-4).compute(n)
    d = mesh.blocks[0].points.data
    d[..., 0] = np.linspace(0.0, 1.0, n)[:, None]
    d[..., 1] = y[None, :]
    with smooth.Smoother(mesh, auto) as sm:
        assert sm.inner == solver.Inner.bicgstab
    # the same block stretched UNIFORMLY (every cell 50 : 1) is what the semi-coarsening is for: the cycle stays
    mesh = configs.single_block(n, n)
    d = mesh.blocks[0].points.data
    d[..., 0] = np.linspace(0.0, 50.0, n)[:, None]
    d[..., 1] = np.linspace(0.0, 1.0, n)[None, :]
    with smooth.Smoother(mesh, auto) as sm:
        assert sm.inner == solver.Inner.mg_bicgstab
    # the T106 example refined 8 x (blocks up to 308 481 nodes, O-grid clustering): plain solve; an explicit choice is kept
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    j = json.load(open(os.path.join(gold, "examples", "T106", "T106.json")))
    nc = j["template"]["O4H"]["num_cells"]
    for k in nc:
        nc[k] *= 2 if k == "o_grid" else 8
    inp = Input.parse(json.dumps(j))
    mesh = inp.template.run(inp.geometry(gold))
    assert max(b.points.size[0] * b.points.size[1] for b in mesh.blocks) >= 100000
    with smooth.Smoother(mesh, auto, inp.wall_control_function) as sm:
        assert sm.inner == solver.Inner.bicgstab
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab), inp.wall_control_function) as sm:
        assert sm.inner == solver.Inner.mg_bicgstab


def test_a_handle_that_never_runs_triples_exchanges_the_depth_2_halo_only(monkeypatch):
    # the depth-3 ghost set (one exchange per sweep TRIPLE) used to be exchanged by every handle on large blocks, Krylov modes
    # included.  Now the depth follows the handle's options -- identical on every rank -- and the library's own transport builds its tables
    # for them (tm_rccl_hooks_for).  Threshold lowered so that these small blocks count as large.
    from turbomesh_amd import distributed as tmd

    monkeypatch.setenv("TM_TRIPLES_MIN_NODES", "1")
    rows = {}
    for name, opt in (("relax", solver.Option.hip(inner=solver.Inner.relax)), ("single_sweep", solver.Option.hip(inner=solver.Inner.relax, single_sweep=True)),
                      ("bicgstab", solver.Option.hip()), ("gmres", solver.Option.hip(inner=solver.Inner.gmres))):
        h = tmd.TorchHooks(tmd.strip_for_rank(3, 1, 40, 50), owner=[0, 1, 2], rank=1, world=3, option=opt)   # rank 1 of 3: neighbours on both sides; never iterated
        rows[name] = (int(sum(h.plan["recv_count"])), int(sum(h.plan["send_count"])))
        h.smoother.close()
    assert rows["bicgstab"] == rows["gmres"] == rows["single_sweep"]
    assert rows["relax"][0] > rows["bicgstab"][0] and rows["relax"][1] > rows["bicgstab"][1], rows
    # a strip: the solved side of an interface sends 3 rows for triples and 2 for pairs, the slaved side 4 and 3
    assert rows["relax"] == (7 * 50, 7 * 50) and rows["bicgstab"] == (5 * 50, 5 * 50), rows


def test_default_schedule_across_ranks_is_triples_at_every_size(monkeypatch):
    # round 4: with the three level passes of a triple in one launch, triples beat pairs down to 256^2 blocks (tools/dev/triples_threshold.sh),
    # so the library's own default -- no TM_TRIPLES_MIN_NODES in the environment -- is the depth-3 halo for every block of at least 16 x 16 nodes
    from turbomesh_amd import distributed as tmd

    monkeypatch.delenv("TM_TRIPLES_MIN_NODES", raising=False)
    relax = solver.Option.hip(inner=solver.Inner.relax)
    h = tmd.TorchHooks(tmd.strip_for_rank(3, 1, 40, 50), owner=[0, 1, 2], rank=1, world=3, option=relax)
    assert int(sum(h.plan["recv_count"])) == 7 * 50
    h.smoother.close()
    h = tmd.TorchHooks(tmd.strip_for_rank(3, 1, 12, 50), owner=[0, 1, 2], rank=1, world=3, option=relax)   # blocks below 16 rows: pairs
    assert int(sum(h.plan["recv_count"])) == 5 * 50
    h.smoother.close()
