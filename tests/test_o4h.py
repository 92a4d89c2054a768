"""SURVEY "next" row N1: the O4H blocking (templates/O4H.py), splines, csv reader and JSON front door, host side.
BASELINE.json configs[0] (T106, CPU plumbing) and configs[2] (LS89, all blocks on one GPU).

CPU: the reference's own KATs for csv / spline; block sizes; the reference-faithful oracle run of the JSON as written.
GPU: TFI of all 8 blocks bit-exact vs the oracle; smoothing (Laplace and `white`) vs the exact-solve oracle."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat, oracle_tfi
from turbomesh_amd import csv as tcsv
from turbomesh_amd import spline
from turbomesh_amd.input import Input
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = {   # SURVEY.md section 8 (derived from O4H.zig:71-74, 129-419 and the two JSONs)
    "T106": [(221, 41), (121, 41), (11, 41), (11, 51), (121, 41), (161, 11), (21, 91), (11, 131)],
    "LS89": [(381, 41), (161, 41), (6, 131), (11, 51), (116, 41), (266, 11), (21, 181), (11, 241)],
}


def load(name, tfi):
    inp = Input.parse(open(os.path.join(GOLD, "examples", name, name + ".json")).read())
    geom = inp.geometry(GOLD)
    return inp, inp.template.run(geom, tfi=tfi)


def test_csv_reference_kat():
    d = tcsv.parseCsvIntoVec2d(os.path.join(GOLD, "examples", "T106", "T106_ps.dat"))   # csv.zig:59-67
    assert d[0, 0] == 1.127030384 and d[0, 1] == -0.047185256
    assert d[-1, 0] == 1.047805900 and d[-1, 1] == 0.000076595


def test_spline_reference_kats():
    s = spline.FittingSpline([[0, 0], [0.5, 0.5], [1, 1], [2, 2], [3, 3], [4, 4]])   # spline.zig:235-262
    v = s.interpolate([0.0, 0.125, 0.25, 0.5, 0.75, 1.0])
    assert np.abs(v - np.array([[0, 0], [0.5, 0.5], [1, 1], [2, 2], [3, 3], [4, 4]])).max() < 1e-9
    assert abs(s.integrate() - np.sqrt(2.0) * 4.0) < 1e-9
    s = spline.FittingSpline([[0, 0], [1, 0.5], [2, 1.5], [2.5, 3.0]])                 # spline.zig:264-287
    v = s.interpolate([0.0, 0.5, 1.0])
    assert v[0, 0] <= v[1, 0] <= v[2, 0]
    assert np.abs(v[0] - [0, 0]).max() < 1e-9 and np.abs(v[2] - [2.5, 3.0]).max() < 1e-9
    assert abs(spline.FittingSpline([[0, 0], [0, 3.0]]).integrate() - 3.0) < 1e-9      # spline.zig:289-302


@pytest.mark.parametrize("name", ["T106", "LS89"])
def test_o4h_block_sizes_and_topology(name):
    inp, mesh = load(name, oracle_tfi)
    assert [b.points.size for b in mesh.blocks] == SIZES[name]
    assert len(mesh.connections) == 21 and sum(c.periodicity is not None for c in mesh.connections) == 3
    assert len(mesh.boundary_conditions) == 2 and inp.iterations == 10
    om = OracleMesh(mesh)
    s = oracle.System(om, ("white", 1e-6, 1.570796327))   # connectionDataCheck (1e-15), row kinds, junctions, CSR pattern
    rows = smooth.plan_rows(mesh)
    assert np.array_equal(rows["kind"], s.boundary_kind)
    assert np.bincount(s.boundary_kind, minlength=5)[oracle.KIND_LAPLACIAN] == 12
    p, ci = s.lhs_p, s.lhs_i
    for k, g in enumerate(rows["row"]):
        assert np.array_equal(ci[p[g]:p[g + 1]], rows["cols"][k, :rows["ncols"][k]])


def test_config1_t106_json_as_written_on_the_cpu_oracle():
    # BASELINE configs[0]: 10 iterations, GMRES + ILU(0), white control function -- the reference's CPU path, restated
    inp, mesh = load("T106", oracle_tfi)
    om = OracleMesh(mesh)
    w = inp.wall_control_function.white
    st = oracle.smooth_mesh(om, inp.iterations, solver=oracle.SOLVER_GMRES, preconditioner=oracle.PRECOND_ILU0, control=("white", w.ds_target, w.theta_target))
    assert st.outer_iterations == 10 and st.not_converged == 0
    assert st.residual_history[-1] < 1e-4 * st.residual_history[0] and np.isfinite(om.flat()).all()


def _t106_block0(tfi):
    """BASELINE configs[0] read literally -- "examples/T106 single 2D block, linear TFI + 100 elliptic iters" (SURVEY 8d config 1,
    variant (ii)): block 0 of the O4H blocking (221 x 41, the O-grid half around the suction side) on its own, all four sides
    fixed, Laplace control function."""
    from turbomesh_amd.discrete import Mesh

    _, full = load("T106", tfi)
    m = Mesh()
    m.addBlock("blade_up", full.blocks[0])
    return m


def test_config1_single_block_100_iterations_on_the_cpu_oracle():
    # 100 Picard iterations on the CPU oracle, three inner solvers against the exact-solve iteration (= the reference with UMFPACK)
    mesh = _t106_block0(oracle_tfi)
    assert mesh.blocks[0].points.data.shape == (221, 41, 2)
    exact = OracleMesh(_t106_block0(oracle_tfi))
    hist, _ = oracle.picard_exact(exact, 100)
    assert hist[-1] < 1e-12 * hist[0]                      # the Picard iteration has converged: (sum dx^2 + sum dy^2)^2

    def run(solver_tag, precond):
        om = OracleMesh(_t106_block0(oracle_tfi))
        st = oracle.smooth_mesh(om, 100, solver=solver_tag, preconditioner=precond)
        assert st.outer_iterations == 100 and st.not_converged == 0
        return st, float(np.sqrt(np.mean((om.flat() - exact.flat()) ** 2)))

    # (1) the scale-aware criterion this build solves with: the exact fixed point
    st, rms = run(oracle.SOLVER_SCALED_BICGSTAB, oracle.PRECOND_DIAGONAL)
    assert rms <= 1e-10, rms
    # (2) the reference's own settings, restated faithfully: the stop test max(1e-8, 1e-6 ||b||) acts on the UNSCALED
    # residual, whose interior rows are O(h^2) -- GMRES + ILU(0) (the solver of T106.json) stops iterating after the first few
    # outer iterations and stalls ~1e-6 RMS short of the fixed point; BiCGStab + diagonal returns before its first iteration
    st, rms = run(oracle.SOLVER_GMRES, oracle.PRECOND_ILU0)
    assert 1e-10 < rms < 5e-6 and st.inner_iterations < 100, (rms, st.inner_iterations)
    st, rms = run(oracle.SOLVER_BICGSTAB, oracle.PRECOND_DIAGONAL)
    assert st.inner_iterations == 0 and not st.residual_history.any() and rms > 1e-5


@pytest.mark.gpu
def test_gpu_config1_single_block_100_iterations():
    # the same job on the GPU through smooth.mesh (one call, 100 Picard iterations): CONVERGED coordinates within 1e-10 RMS of
    # the CPU oracle's (north_star), TFI seed bit-identical
    mesh = _t106_block0(None)
    ref = _t106_block0(oracle_tfi)
    assert mesh.blocks[0].points.data.tobytes() == ref.blocks[0].points.data.tobytes()
    exact = OracleMesh(ref)
    hist, _ = oracle.picard_exact(exact, 100)
    x0 = mesh.blocks[0].points.data.copy()
    st = smooth.mesh(mesh, 100, solver.Option.hip(rtol=1e-13, max_inner=20000))
    assert st["outer_iterations"] == 100 and st["not_converged"] == 0
    rms = float(np.sqrt(np.mean((mesh_flat(mesh) - exact.flat()) ** 2)))
    moved = float(np.sqrt(np.mean((mesh_flat(mesh) - x0.reshape(-1, 2)) ** 2)))
    assert rms <= 1e-10, rms
    assert moved > 1e-5                                    # the smoothing did something: the TFI seed is 7e-5 RMS from the fixed point
    assert st["last_residual"] <= 1e-12 * hist[0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["T106", "LS89"])
def test_gpu_o4h_tfi_and_smoothing(name):
    inp, mesh = load(name, None)              # Block2d.init -> TFI on the GPU
    _, mesh_cpu = load(name, oracle_tfi)
    for a, b in zip(mesh.blocks, mesh_cpu.blocks):
        assert a.points.data.tobytes() == b.points.data.tobytes()
    # Laplace control function: 2 Picard iterations vs the exact-solve oracle
    om = OracleMesh(mesh)
    hist, _ = oracle.picard_exact(om, 2)
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-13, max_inner=20000)) as sm:
        st = sm.iterate(2)
        sm.download()
    assert st["not_converged"] == 0
    rms = float(np.sqrt(np.mean((mesh_flat(mesh) - om.flat()) ** 2)))
    assert rms <= 1e-10, rms
    assert st["last_residual"] == pytest.approx(hist[-1], rel=1e-5)


@pytest.mark.gpu
def test_gpu_t106_white_control_function():
    inp, mesh = load("T106", None)
    w = inp.wall_control_function.white
    om = OracleMesh(mesh)
    oracle.picard_exact(om, 3, control=("white", w.ds_target, w.theta_target))
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-13, max_inner=20000), wcf.Algorithm(wcf.White(w.ds_target, w.theta_target))) as sm:
        st = sm.iterate(3)
        sm.download()
    rms = float(np.sqrt(np.mean((mesh_flat(mesh) - om.flat()) ** 2)))
    assert rms <= 1e-10, (rms, st)   # north_star's bar; acos / atan2 of (P,Q) are the reference's algorithm on both sides (achieved: 6e-12)


@pytest.mark.gpu
def test_cli_runs_the_t106_input_with_the_hip_solver(tmp_path):
    # python -m turbomesh_amd <json> = gui/main.zig:27-56 without the GUI; the file's own solver (gmres) is refused like a
    # reference build without the external solver, --hip swaps in the device solver
    import subprocess
    import sys

    from turbomesh_amd import output

    cfg = os.path.join("examples", "T106", "T106.json")   # like the reference: paths relative to the working directory
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-m", "turbomesh_amd", cfg], capture_output=True, text=True, timeout=300, cwd=GOLD, env=env)
    assert r.returncode != 0 and "ExternalSolverNotEnabled" in r.stderr
    out = str(tmp_path / "t106.xyz")
    r = subprocess.run([sys.executable, "-m", "turbomesh_amd", cfg, "--hip", "--iterations", "2", "--output", out], capture_output=True, text=True,
                       timeout=300, cwd=GOLD, env=env)
    assert r.returncode == 0, r.stderr
    # --hip without a strategy = auto: on these blocks (<= 9061 nodes) that is the plain Picard + BiCGStab solve (7x faster here than the
    # multigrid-preconditioned one); the program says which one it took
    assert "inner strategy: bicgstab (chosen from the block sizes" in r.stderr, r.stderr[-2000:]
    blocks = output.read_plot3d(out)
    assert [(b[0], b[1]) for b in blocks] == [(221, 41), (121, 41), (11, 41), (11, 51), (121, 41), (161, 11), (21, 91), (11, 131)]
    # --hip file: the solver the JSON names ("gmres" + "ilu0", examples/T106/T106.json:28-41) honoured by its device counterpart -- GMRES(30) as
    # GMRES.zig:300-423 with the diagonal in ILU(0)'s place -- and the result is the same mesh (a Krylov method changes the route only)
    out2 = str(tmp_path / "t106_gmres.xyz")
    r = subprocess.run([sys.executable, "-m", "turbomesh_amd", cfg, "--hip", "file", "--iterations", "2", "--output", out2], capture_output=True, text=True,
                       timeout=600, cwd=GOLD, env=env)
    assert r.returncode == 0, r.stderr
    assert "inner strategy: gmres" in r.stderr and "ilu0 has no device counterpart" in r.stderr, r.stderr[-2000:]
    blocks2 = output.read_plot3d(out2)
    for a, b in zip(blocks, blocks2):
        assert np.sqrt(np.mean((a[2] - b[2]) ** 2 + (a[3] - b[3]) ** 2)) <= 1e-10
    assert all(np.isfinite(b[2]).all() and np.isfinite(b[3]).all() for b in blocks)
