"""Picard parity up the size ladder: is "every Picard iterate within 1e-10 RMS of the exact-solve iterate" still
true where the conditioning of the frozen system (~ n^2) bites?

Each case: a perturbed block (interior nodes displaced by 0.25 h, SURVEY 8d config 2's convergence variant) or a roughened 2-block
strip, TWO Picard iterates (smooth.zig:104-154) through the GPU's `bicgstab` (the reference's inner solver, BiCGStab.zig:279-370, on
D^-1 A, matrix-free) and `mg_bicgstab` with the library's DEFAULT options, against the reference run with an exact linear solve
(its UMFPACK backend, umfpack.zig:18-24 = oracle-assembled CSR + scipy splu).  Beside it the oracle's distance from ITSELF with a
second elimination order -- the fp64 floor of "the exact iterate" at that size.  Every figure goes to gpurun_out/parity_rms.log;
the table in DESIGN.md section 5 is that log.  Measured on the MI355X box (round 3):

      n     floor (splu vs splu)   bicgstab, rtol 1e-14   mg_bicgstab
     129         8e-15                  5e-13                9e-15
     257         1e-14                  7e-12                2e-14
     513         5e-14                  2e-11                5e-14
    1025         2e-13                  2e-11                4e-13
"""
import time

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from tests.test_gpu_benchsize import log_parity
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu
TOL_RMS = 1e-10   # BASELINE.json north_star, fp64


def _rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


def _flat(iterates):
    return [np.concatenate([b.reshape(-1, 2) for b in it], axis=0) for it in iterates]


def _rough_strip(n):
    m = configs.strip(2, n, n)
    rng = np.random.default_rng(7)
    for b in m.blocks:   # interior nodes only: the two copies of the interface stay matched (connectionDataCheck, smooth.zig:220-275)
        d = b.points.data
        d[1:-1, 1:-1] += 0.25 / (n - 1) * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
    return m


CASES = {
    "block257": lambda: configs.single_block(257, 257, perturb=0.25),
    "block513": lambda: configs.single_block(513, 513, perturb=0.25),
    "block1025": lambda: configs.single_block(1025, 1025, perturb=0.25),
    "strip2x513": lambda: _rough_strip(513),
}


@pytest.mark.parametrize("name", list(CASES))
def test_two_picard_iterates_vs_exact_solve(name):
    build = CASES[name]
    seed = build()
    t0 = time.time()
    ref = _flat(oracle.picard_exact(OracleMesh(seed), 2, keep_iterates=True)[1])
    alt = _flat(oracle.picard_exact(OracleMesh(seed), 2, keep_iterates=True, permc_spec="MMD_AT_PLUS_A")[1])
    floor = max(_rms(a, b) for a, b in zip(ref, alt))
    log_parity(f"ladder_{name}_oracle_floor", floor)
    print(f"[ladder] {name}: two sparse LUs in {time.time() - t0:.0f} s, floor {floor:.1e}; first step moves {_rms(ref[0], mesh_flat(seed)):.1e}")
    assert floor <= 1e-11, floor      # the yardstick itself is good to two orders below the bar at these sizes
    for inner in (solver.Inner.bicgstab, solver.Inner.mg_bicgstab):
        mesh = build()
        worst = 0.0
        with smooth.Smoother(mesh, solver.Option.hip(inner=inner, max_inner=100000)) as sm:   # default rtol (1e-14), default everything
            for k in range(2):
                st = sm.iterate(1)
                sm.download()
                assert st["not_converged"] == 0, (name, inner.name, st)
                worst = max(worst, _rms(mesh_flat(mesh), ref[k]))
        log_parity(f"ladder_{name}_{inner.name}_rms", worst)
        assert worst <= TOL_RMS, (name, inner.name, worst)
        if inner == solver.Inner.mg_bicgstab:   # a preconditioner that makes the conditioning O(1) lands on the floor itself
            # (a lone block: O(1) indeed.  On coupled blocks the cycle is block-local, and the perimeter treatment around it -- Dirichlet data,
            # perimeter rows, a Jacobi sweep on the perimeter system: Smoother::precondition -- leaves the smooth interface modes to the Krylov
            # iteration: the same residual tolerance then stops a few times further from the exact iterate, 3.4e-12 on this strip)
            assert worst <= (max(20.0 * floor, 1e-12) if len(seed.blocks) == 1 else 1e-11), (name, worst, floor)


def test_default_tolerance_is_size_aware():
    # csrc/tm_smoother.hpp default_rtol / default_max_inner: 1e-14 up to 866^2 nodes, then 7.5e-9 / nodes, never below 1e-16 --
    # observable through the iteration counts of the same solve with rtol = 0 and with the value spelled out
    n = 1400
    a, b = configs.single_block(n, n, perturb=0.25), configs.single_block(n, n, perturb=0.25)
    with smooth.Smoother(a) as sm:
        sa = sm.iterate(1)
        sm.download()
    with smooth.Smoother(b, solver.Option.hip(rtol=7.5e-9 / (n * n), max_inner=int(12 * n))) as sm:
        sb = sm.iterate(1)
        sm.download()
    assert sa["inner_iterations"] == sb["inner_iterations"] and sa["not_converged"] == 0
    assert np.array_equal(a.blocks[0].points.data, b.blocks[0].points.data)


def test_4096_two_routes_to_the_same_picard_iterates():
    # BASELINE configs[1] at its full size, the reference's algorithm (Picard + linear solve, smooth.zig:104-154).  No sparse LU
    # holds 16.7 M unknowns, so the two GPU routes check each other: `mg_bicgstab` (8 inner iterations per solve) against
    # `bicgstab` with the diagonal alone (~18 000), both with DEFAULT options.  They share the operator kernels and the Krylov
    # recurrences but nothing of what decides the error -- the spectrum each leaves in its final residual -- and each is tied to
    # the sparse-LU oracle up to 2049^2 (1025^2 in the test above, 2049^2: tools/dev/picard_ladder.py).
    # Fixed boundary bit-exact; the Picard update decreases.
    n = 4096
    seed = configs.single_block(n, n, perturb=0.25)
    x0 = seed.blocks[0].points.data.copy()
    del seed
    out = {}
    for inner, iters in ((solver.Inner.mg_bicgstab, 3), (solver.Inner.bicgstab, 1)):
        mesh = configs.single_block(n, n, perturb=0.25)
        its, upd = [], []
        with smooth.Smoother(mesh, solver.Option.hip(inner=inner)) as sm:
            for k in range(iters):
                st = sm.iterate(1)
                sm.download()
                assert st["not_converged"] == 0, (inner.name, st)
                its.append(mesh.blocks[0].points.data.copy())
                upd.append(float(np.sqrt((st["last_dx2"] + st["last_dy2"]) / (n * n))))
                print(f"[4096] {inner.name} iterate {k + 1}: inner {st['inner_iterations']}, update rms {upd[-1]:.2e}, {st['seconds']:.2f} s")
        out[inner] = (its, upd)
        d = its[-1]
        assert np.array_equal(d[0], x0[0]) and np.array_equal(d[-1], x0[-1]) and np.array_equal(d[:, 0], x0[:, 0]) and np.array_equal(d[:, -1], x0[:, -1])
    mg_its, mg_upd = out[solver.Inner.mg_bicgstab]
    assert mg_upd[0] > mg_upd[1] > mg_upd[2], mg_upd
    rms = _rms(mg_its[0], out[solver.Inner.bicgstab][0][0])
    log_parity("ladder_block4096_mg_vs_bicgstab_default_options_rms", rms)
    assert rms <= TOL_RMS, rms
    # ... and against something that shares NO code with the device: the reference's system at X^0 assembled by the faithful oracle on the host
    # (smooth.zig:309-385, 923-1113 restated: CSR + right-hand sides, 151 M non-zeros), the TRUE scaled residual of each route's X^1 formed
    # in 80-bit arithmetic (tests/residual_check.py).  No fp64 vector can read below the storage floor 2^-53 ||D^-1 |A| |x||| / ||D^-1 b||
    # (6e-15 here -- above the 4.5e-16 the recurrence residual is driven to): the multigrid route sits ON it, the diagonal-only route stalls
    # ~40 x above it (its remaining error is low-frequency: little residual, and the part that the tight recurrence tolerance keeps removing).
    from tests import residual_check as rc

    p, ci, v, b = rc.assemble([x0])
    assert len(p) - 1 == n * n
    for inner, bound in ((solver.Inner.mg_bicgstab, 4.0), (solver.Inner.bicgstab, 200.0)):
        x1 = out[inner][0][0].reshape(-1, 2)
        both, per, floor = rc.relative_residual(p, ci, v, b, x1, with_floor=True)
        log_parity(f"block4096_{inner.name}_true_scaled_residual_vs_oracle_assembled_system", both)
        log_parity(f"block4096_{inner.name}_true_scaled_residual_over_fp64_storage_floor", both / floor)
        print(f"[4096] {inner.name}: true scaled residual vs the oracle-assembled system {both:.2e} = {both / floor:.1f} x the fp64 storage floor {floor:.2e}")
        assert both <= bound * floor and both <= 1e-12, (inner.name, both, floor)
    # ... and the OPERATOR itself at this size, row by row, against the reference-order arithmetic: A(X^0) w through the matrix-free kernels
    # (tm_smoother_apply) vs the oracle-assembled CSR times w (scipy: the sum of the nine products in CSR order, i.e. BiCGStab.zig:424-435) --
    # perimeter rows bit for bit, interior rows within 16 eps sum_k |c_k w_k| (the factored row evaluation; tests/test_gpu_operator.py holds
    # the same bound on the small topologies, here on all 16.7 M rows)
    import scipy.sparse as sp

    A = sp.csr_matrix((v, ci, p), shape=(n * n, n * n))
    w = np.random.default_rng(11).standard_normal((n * n, 2))
    blk = configs.block_from_array(x0.copy())
    from turbomesh_amd.discrete import Mesh

    m0 = Mesh()
    m0.addBlock("b", blk)
    with smooth.Smoother(m0, solver.Option.hip()) as sm:
        got = sm.apply(w, scaled=False)
        kinds = sm.row_kinds()
        # the device's OWN assembly of the reference's system (tm_smoother_assemble_csr: StencilData.init in reference order on every one of the
        # 16.7 M rows) against the oracle's arrays: pattern and all 151 M coefficients bit for bit
        Ap_d, Ai_d, Ax_d, _ = sm.assemble_csr()
        assert np.array_equal(Ap_d, p) and np.array_equal(Ai_d, ci)
        assert np.array_equal(Ax_d, v), f"{np.count_nonzero(Ax_d != v)} of {len(v)} coefficients differ"
        del Ap_d, Ai_d, Ax_d
    ref = np.stack([A @ w[:, 0], A @ w[:, 1]], axis=1)
    perim = kinds >= 0
    assert np.array_equal(got[perim], ref[perim])
    absA = abs(A)
    bnd = 16 * np.finfo(float).eps * np.stack([absA @ np.abs(w[:, 0]), absA @ np.abs(w[:, 1])], axis=1)
    worst = float((np.abs(got - ref) / (bnd + 1e-300)).max())
    log_parity("block4096_operator_vs_reference_order_csr_worst_over_16eps_bound", worst)
    assert worst <= 1.0, worst


def test_1025_three_routes_one_of_them_through_the_oracle_assembled_csr():
    # the same statement with a THIRD route that shares no operator kernel with the other two: the oracle's CSR itself solved on the device
    # through seam 2 (tm_csr_solve: the CSR mat-vec of csrc/tm_csr.hip on the reference-order coefficients) from the same initial guess.
    # All three X^1 within 1e-10 rms of one another and every true residual within a small multiple of the storage floor.
    # (4096^2: tools/oracle_residual.py 4096 --csr, 102 s for the CSR route -- profiles/r04_parity_rms.txt: 4.9e-12 / 1.4e-11.)
    import ctypes as C

    from tests import residual_check as rc
    from turbomesh_amd import _capi

    n = 1025
    x0 = configs.single_block(n, n, perturb=0.25).blocks[0].points.data.copy()
    its = {}
    for inner in (solver.Inner.mg_bicgstab, solver.Inner.bicgstab, solver.Inner.gmres):
        m = configs.single_block(n, n, perturb=0.25)
        with smooth.Smoother(m, solver.Option.hip(inner=inner, max_inner=200000)) as sm:
            st = sm.iterate(1)
            sm.download()
        assert st["not_converged"] == 0, (inner.name, st)
        its[inner.name] = m.blocks[0].points.data.reshape(-1, 2).copy()
    p, ci, v, b = rc.assemble([x0])
    xs, ys = np.ascontiguousarray(x0.reshape(-1, 2)[:, 0]).copy(), np.ascontiguousarray(x0.reshape(-1, 2)[:, 1]).copy()
    bx, by = np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])
    opt = solver.Option.hip(rtol=7.5e-9 / (n * n), max_inner=200000).c_struct()   # the matrix-free path's size-aware default, spelled out (seam 2 defaults to 1e-14)
    st = _capi.tm_stats()
    ip = C.POINTER(C.c_int32)
    rcode = _capi.lib().tm_csr_solve(len(p) - 1, p.ctypes.data_as(ip), ci.ctypes.data_as(ip), _capi.f64ptr(v), None, _capi.f64ptr(bx), _capi.f64ptr(by),
                                     _capi.f64ptr(xs), _capi.f64ptr(ys), C.byref(opt), C.byref(st))
    assert rcode == 0, rcode
    its["csr_seam2"] = np.stack([xs, ys], axis=1)
    names = list(its)
    for a in range(len(names)):
        for c in range(a + 1, len(names)):
            d = _rms(its[names[a]], its[names[c]])
            log_parity(f"block1025_{names[a]}_vs_{names[c]}_rms", d)
            assert d <= TOL_RMS, (names[a], names[c], d)
    for name, x1 in its.items():
        both, _, floor = rc.relative_residual(p, ci, v, b, x1, with_floor=True)
        log_parity(f"block1025_{name}_true_scaled_residual_over_fp64_storage_floor", both / floor)
        assert both <= 200.0 * floor, (name, both, floor)


def test_1025_both_routes_reach_the_same_fixed_point():
    # Picard run to convergence of the COORDINATES -- the update of the last iteration, the reference's own per-iteration quantity
    # (smooth.zig:112-137), <= 1e-11 RMS -- through both inner solvers: same fixed point to 1e-10 RMS
    n = 1025
    res = []
    for inner in (solver.Inner.mg_bicgstab, solver.Inner.bicgstab):
        mesh = configs.single_block(n, n, perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=inner)) as sm:
            reached, st = sm.iterate_until_update(1e-11, 60)
            sm.download()
        assert reached, st
        print(f"[1025 fixed point] {inner.name}: {st['outer_iterations']} Picard iterations, {st['inner_iterations']} inner, {st['seconds']:.2f} s")
        res.append(mesh.blocks[0].points.data.copy())
    rms = _rms(res[0], res[1])
    log_parity("ladder_block1025_fixed_point_mg_vs_bicgstab_rms", rms)
    assert rms <= TOL_RMS, rms


@pytest.mark.parametrize("name", ["T106", "LS89"])
def test_o4h_examples_at_four_times_the_nodes(name):
    # Every row kind of the reference at scale -- interfaces, 3 periodic connections, 12 junction points, sliding inlet / outlet rows,
    # Roberts boundary-layer clustering, the White control function of the JSON -- on the example meshes with every num_cells entry
    # doubled (T106: 98 228 nodes, LS89: 147 398; x3 / x4 leave the domain of the reference's tanh clustering formula,
    # clustering.zig:56-95).  Two Picard iterates, default options, against the sparse-LU oracle (tools/dev/o4h_refined_probe.py
    # also runs the multigrid route: 5e-13 / 9e-14).
    import json
    import os

    from tests.test_o4h import GOLD
    from turbomesh_amd.input import Input
    from turbomesh_amd.smoothing import wall_control_function as wcf

    j = json.load(open(os.path.join(GOLD, "examples", name, name + ".json")))
    nc = j["template"]["O4H"]["num_cells"]
    for k in nc:
        nc[k] *= 2
    inp = Input.parse(json.dumps(j))
    build = lambda: inp.template.run(inp.geometry(GOLD), tfi=None)
    mesh = build()
    w = inp.wall_control_function.white
    ref = _flat(oracle.picard_exact(OracleMesh(mesh), 2, control=("white", w.ds_target, w.theta_target), keep_iterates=True)[1])
    worst = 0.0
    with smooth.Smoother(mesh, solver.Option.hip(), wcf.Algorithm(wcf.White(w.ds_target, w.theta_target))) as sm:
        for k in range(2):
            st = sm.iterate(1)
            sm.download()
            assert st["not_converged"] == 0, st
            worst = max(worst, _rms(mesh_flat(mesh), ref[k]))
    log_parity(f"ladder_{name}_x2_white_bicgstab_rms", worst)
    assert worst <= TOL_RMS, worst
