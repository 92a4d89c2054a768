"""CPU: the product's host-side planning (turbomesh_amd/csrc/tm_plan.cpp, exported by tm_plan_build) against the
oracle's assembled CSR rows (reference smooth.zig:421-921): kinds, column pattern, static coefficients,
right-hand sides, and the stencil-slot wiring of `smoothed` interface rows.  No GPU needed."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from tests.meshes import TOPOLOGIES
from turbomesh_amd import TmError, configs
from turbomesh_amd.boundary import Condition, ConditionTag, Connection, Range, Side
from turbomesh_amd.smoothing import smooth

SLOT_NAMES = ["i_j", "ip1_j", "im1_j", "i_jp1", "i_jm1", "ip1_jp1", "ip1_jm1", "im1_jp1", "im1_jm1"]   # smooth.zig:175-185


def _stencil(im1_j, ip1_j, i_jm1, i_jp1, P=0.0, Q=0.0):
    """StencilData.init (smooth.zig:192-215) in numpy float64, same expression order."""
    x_xi = 0.5 * (ip1_j[0] - im1_j[0])
    x_eta = 0.5 * (i_jp1[0] - i_jm1[0])
    y_xi = 0.5 * (ip1_j[1] - im1_j[1])
    y_eta = 0.5 * (i_jp1[1] - i_jm1[1])
    g22 = x_eta * x_eta + y_eta * y_eta
    g12 = x_xi * x_eta + y_xi * y_eta
    g11 = x_xi * x_xi + y_xi * y_xi
    return [-2.0 * g22 - 2.0 * g11, g22 * (1 + 0.5 * P), g22 * (1 - 0.5 * P), g11 * (1 + 0.5 * Q), g11 * (1 - 0.5 * Q), -0.5 * g12, 0.5 * g12,
            0.5 * g12, -0.5 * g12]


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_plan_rows_equal_reference_assembly(name):
    mesh = TOPOLOGIES[name](oracle_tfi)
    om = OracleMesh(mesh)
    s = oracle.System(om)
    s.fill(0)
    rows = smooth.plan_rows(mesh)
    p, ci = s.lhs_p, s.lhs_i
    kinds = s.boundary_kind
    assert len(rows["row"]) == len(kinds)
    assert np.array_equal(rows["kind"], kinds)           # perimeter order == ascending global id
    assert np.all(np.diff(rows["row"]) > 0)
    flat = om.flat()
    s.fill_x_specific()
    vx = s.lhs_values.copy()
    bx = s.rhs_x.copy()
    s.fill_y_specific()
    vy = s.lhs_values.copy()
    by = s.rhs_y.copy()
    for k, g in enumerate(rows["row"]):
        nc = rows["ncols"][k]
        assert np.array_equal(ci[p[g]:p[g + 1]], rows["cols"][k, :nc]), (name, g)
        if rows["kind"][k] == oracle.KIND_SMOOTHED:
            # coefficients are dynamic: rebuild them from the slot wiring and compare with the assembled values
            cols = rows["cols"][k]
            slot = rows["slot"][k]
            where = {SLOT_NAMES[slot[q]]: cols[q] for q in range(9)}
            assert where["i_j"] == g and len(set(where.values())) == 9
            vals = vx[p[g]:p[g + 1]]
            # the four metric neighbours are the columns wired to im1_j, ip1_j, i_jm1, i_jp1
            per = np.zeros(2)
            if not np.all(bx[g] == 0) or not np.all(by[g] == 0):
                pass
            conn_per = [c.periodicity for c in mesh.connections if c.periodicity is not None]
            i_jp1 = flat[where["i_jp1"]].copy()
            if conn_per and name.startswith("channel"):
                per = np.array(conn_per[0])
                i_jp1 = i_jp1 + (-per)
            c9 = _stencil(flat[where["im1_j"]], flat[where["ip1_j"]], flat[where["i_jm1"]], i_jp1)
            for q in range(9):
                assert vals[q] == c9[slot[q]], (name, g, q)
            if per.any():
                cs = c9[7] + c9[3] + c9[5]
                assert bx[g] == per[0] * cs and by[g] == per[1] * cs
        else:
            assert np.array_equal(vx[p[g]:p[g + 1]], rows["coef_x"][k, :nc]), (name, g, rows["kind"][k])
            assert np.array_equal(vy[p[g]:p[g + 1]], rows["coef_y"][k, :nc]), (name, g, rows["kind"][k])
            for comp, b in ((0, bx), (1, by)):
                want = rows["rhs"][k, comp]
                if np.isnan(want):          # taken from the coordinates (fixed rows; x of sliding rows)
                    assert b[g] == flat[g, comp]
                else:
                    assert b[g] == want


def test_plan_full_size_counts():
    # BASELINE sizes: the table is perimeter-sized (never 9*dof like the reference's CSR, smooth.zig:320-352)
    mesh = configs.strip(8, 2048, 2048, tfi=lambda *e: configs.block_from_array(np.zeros((2048, 2048, 2))))
    rows = smooth.plan_rows(mesh)
    assert len(rows["row"]) == 8 * 2 * (2048 + 2048 - 2)
    k = np.bincount(rows["kind"], minlength=5)
    assert k[oracle.KIND_SMOOTHED] == 7 * 2046 and k[oracle.KIND_CONNECTED] == 7 * 2048 and k[oracle.KIND_LAPLACIAN] == 0


def _mesh2():
    return configs.strip(2, 9, 12, tfi=oracle_tfi)


def test_plan_topology_errors():
    m = _mesh2()
    m.connections[0] = Connection((Range(1, Side.j_min, 0, 11), Range(0, Side.j_max, 0, 11)), None)   # block order (smooth.zig:562)
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -2
    m = _mesh2()
    m.connections[0] = Connection((Range(0, Side.j_max, 0, 4), Range(1, Side.j_min, 0, 4)), None)     # lenInternal() > 3 (smooth.zig:631)
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -2
    m = _mesh2()
    m.connections[0] = Connection((Range(0, Side.j_max, 0, 11), Range(1, Side.j_min, 0, 10)), None)   # unequal lengths
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -3
    m = _mesh2()
    m.connections[0] = Connection((Range(0, Side.j_max, 0, 12), Range(1, Side.j_min, 0, 12)), None)   # beyond the side
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -2
    m = _mesh2()
    m.boundary_conditions.append(Condition(Range(0, Side.j_min, 0, 11), ConditionTag.wall))            # smooth.zig:775 unreachable
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -2
    m = configs.single_block(9, 9, tfi=oracle_tfi)
    m.connections.append(Connection((Range(0, Side.j_min, 0, 8), Range(0, Side.j_max, 0, 8)), None))   # same block only i_min->i_max
    with pytest.raises(TmError) as ei:
        smooth.plan_rows(m)
    assert ei.value.code == -2
    # the oracle rejects the same meshes
    with pytest.raises(oracle.OracleError):
        oracle.System(OracleMesh(m))
