"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle + scipy splu):
CPU: the oracle's own direct solver reproduces them; GPU: the HIP path matches them (TFI bit-exact,
Picard iterates <= 1e-10 RMS, BASELINE.json tolerance)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.boundary import Condition, ConditionTag, Connection, Range, Side
from turbomesh_amd.discrete import Edge, Mesh

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMOOTH = sorted(glob.glob(os.path.join(HERE, "smooth_*.npz")))
TFI = sorted(glob.glob(os.path.join(HERE, "tfi_*.npz")))


def load_mesh(z):
    m = Mesh()
    for b in range(int(z["nblocks"])):
        m.addBlock(f"b{b}", configs.block_from_array(z[f"seed_{b}"].copy()))
    for row, per in zip(z["conns"], z["periodicity"]):
        r0 = Range(int(row[0]), Side(int(row[1])), int(row[2]), int(row[3]))
        r1 = Range(int(row[4]), Side(int(row[5])), int(row[6]), int(row[7]))
        m.connections.append(Connection((r0, r1), None if np.isnan(per[0]) else (float(per[0]), float(per[1]))))
    for row in z["bcs"]:
        m.boundary_conditions.append(Condition(Range(int(row[0]), Side(int(row[1])), int(row[2]), int(row[3])), ConditionTag(int(row[4]))))
    ctl = z["control"]
    control = None if ctl[0] == 0 else ("white", float(ctl[1]), float(ctl[2]))
    return m, control


def test_fixtures_present():
    assert len(SMOOTH) >= 6 and len(TFI) >= 2


@pytest.mark.parametrize("path", TFI, ids=[os.path.basename(p) for p in TFI])
def test_oracle_tfi_matches_golden(path):
    z = np.load(path)
    out = oracle.tfi_block(z["x_i_min"], z["x_i_max"], z["x_j_min"], z["x_j_max"], z["s1"], z["s2"], z["t1"], z["t2"])
    assert out.tobytes() == z["field"].tobytes()


@pytest.mark.parametrize("path", SMOOTH, ids=[os.path.basename(p) for p in SMOOTH])
def test_oracle_direct_solver_matches_golden(path):
    z = np.load(path)
    mesh, control = load_mesh(z)
    n_it = len(z["residual_history"])
    for k in range(n_it):
        om = OracleMesh(mesh)
        st = oracle.smooth_mesh(om, k + 1, solver=oracle.SOLVER_DIRECT, control=control)
        ref = np.concatenate([z[f"iter{k + 1}_{b}"].reshape(-1, 2) for b in range(len(mesh.blocks))])
        assert np.abs(om.flat() - ref).max() < 1e-11
        assert st.last_residual == pytest.approx(z["residual_history"][k], rel=1e-6, abs=1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("path", TFI, ids=[os.path.basename(p) for p in TFI])
def test_hip_tfi_matches_golden(path):
    from turbomesh_amd.discrete import Block2d

    z = np.load(path)
    blk = Block2d.init(Edge(z["x_i_min"], z["s1"]), Edge(z["x_i_max"], z["s2"]), Edge(z["x_j_min"], z["t1"]), Edge(z["x_j_max"], z["t2"]))
    assert blk.points.data.tobytes() == z["field"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("path", SMOOTH, ids=[os.path.basename(p) for p in SMOOTH])
def test_hip_smoothing_matches_golden(path):
    from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

    z = np.load(path)
    mesh, control = load_mesh(z)
    algo = wcf.Algorithm.laplace() if control is None else wcf.Algorithm(wcf.White(control[1], control[2]))
    tol = 1e-10   # north_star's bar, also with the white control function (acos / atan2 of (P,Q): the reference's algorithm in the fixtures' generator and on the device)
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-13, max_inner=5000), algo) as sm:
        for k in range(len(z["residual_history"])):
            st = sm.iterate(1)
            sm.download()
            ref = np.concatenate([z[f"iter{k + 1}_{b}"].reshape(-1, 2) for b in range(len(mesh.blocks))])
            rms = float(np.sqrt(np.mean((mesh_flat(mesh) - ref) ** 2)))
            assert rms <= tol, (os.path.basename(path), k, rms)
            assert st["last_residual"] == pytest.approx(z["residual_history"][k], rel=1e-5, abs=1e-30)
