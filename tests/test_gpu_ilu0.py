"""ILU(0) on the device: the reference's second preconditioner (preconditioner.zig:1-4; factorisation BiCGStab.zig:178-277,
application :384-422, identical copies in GMRES.zig:199-298, 437-475) in the linear-solver slot (tm_csr_solve, TM_OPT_PRECOND_ILU0).

The recurrence is sequential row by row; the device runs it by the LEVEL SETS of its dependency graph with every row in the reference's own
operation order -- so the factor and M^-1 r must equal the faithful CPU restatement (oracle/orc_solvers.cpp) BIT FOR BIT, on the systems
the oracle assembles from the test topologies (junctions, periodic + sliding rows, reversed interfaces) and on the reference's 5 x 5 known
answer with its two missing diagonals.  As preconditioner of the device BiCGStab it cuts the iteration count several-fold and lands on the
same solution."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from tests.meshes import TOPOLOGIES
from tests.test_gpu_csr_solve import csr_solve
from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu
_ip = C.POINTER(C.c_int32)


def device_ilu0(Ap, Ai, Ax, rhs):
    Ap = np.ascontiguousarray(Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Ax = np.ascontiguousarray(Ax, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    n = len(Ap) - 1
    lu, z = np.empty(len(Ax)), np.empty(n)
    _capi.check(_capi.lib().tm_csr_ilu0_probe(n, Ap.ctypes.data_as(_ip), Ai.ctypes.data_as(_ip), _capi.f64ptr(Ax), _capi.f64ptr(rhs), _capi.f64ptr(lu), _capi.f64ptr(z)))
    return lu, z


def _system(name):
    om = OracleMesh(TOPOLOGIES[name](oracle_tfi))
    s = oracle.System(om)
    s.fill(0)
    return om, s


@pytest.mark.parametrize("name", ["single_perturbed_33", "strip3_reversed", "channel_periodic_sliding", "two_by_two_junction", "strip2_40x300", "plate_le"])
def test_factor_and_application_equal_the_reference_recurrence_bit_for_bit(name):
    om, s = _system(name)
    s.fill_x_specific()
    p, ci, v = s.lhs_p.copy(), s.lhs_i.copy(), s.lhs_values.copy()
    rhs = np.random.default_rng(7).standard_normal(len(p) - 1)
    lu_ref, z_ref = oracle.csr_ilu0(len(p) - 1, p, ci, v, rhs)
    lu, z = device_ilu0(p, ci, v, rhs)
    assert np.array_equal(lu, lu_ref), f"{name}: factor differs in {np.count_nonzero(lu != lu_ref)} of {len(lu)} entries"
    assert np.array_equal(z, z_ref), f"{name}: M^-1 r differs, max {np.abs(z - z_ref).max():.2e}"
    assert np.isfinite(z).all() and not np.array_equal(lu, v)


def test_missing_and_zero_diagonals_count_as_one():
    # umfpack.zig:71-97's matrix has rows without a diagonal entry; the reference treats a missing or zero pivot as 1.0 (BiCGStab.zig:240-249, 413-418)
    import scipy.sparse as sp

    A = sp.csc_matrix(([2.0, 3.0, 3.0, -1.0, 4.0, 4.0, -3.0, 1.0, 2.0, 2.0, 6.0, 1.0], [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4], [0, 2, 5, 9, 10, 12]), shape=(5, 5)).tocsr()
    A.sort_indices()
    rhs = np.array([8.0, 45.0, -3.0, 3.0, 19.0])
    lu_ref, z_ref = oracle.csr_ilu0(5, A.indptr, A.indices, A.data, rhs)
    lu, z = device_ilu0(A.indptr, A.indices, A.data, rhs)
    assert np.array_equal(lu, lu_ref) and np.array_equal(z, z_ref)
    # an explicit ZERO pivot (rows 0 and 2 store a 0.0 on the diagonal)
    B = sp.csr_matrix(([0.0, 1.0, 1.0, 2.0, 1.0, 1.0, 0.0], [0, 1, 0, 1, 2, 1, 2], [0, 2, 5, 7]), shape=(3, 3))
    r3 = np.array([1.0, 2.0, 3.0])
    lu_ref, z_ref = oracle.csr_ilu0(3, B.indptr, B.indices, B.data, r3)
    lu, z = device_ilu0(B.indptr, B.indices, B.data, r3)
    assert np.array_equal(lu, lu_ref) and np.array_equal(z, z_ref)


def test_wide_levels_take_the_multi_workgroup_path():
    # a block-diagonal system of 3000 independent 2 x 2 blocks: two levels of 3000 rows each (> 256: one launch of twelve workgroups per level)
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    blocks = [np.array([[4.0 + rng.random(), rng.random()], [rng.random(), 5.0 + rng.random()]]) for _ in range(3000)]
    A = sp.block_diag(blocks, format="csr")
    A.sort_indices()
    rhs = rng.standard_normal(6000)
    lu_ref, z_ref = oracle.csr_ilu0(6000, A.indptr, A.indices, A.data, rhs)
    lu, z = device_ilu0(A.indptr, A.indices, A.data, rhs)
    assert np.array_equal(lu, lu_ref) and np.array_equal(z, z_ref)
    assert np.allclose(A @ z, rhs, rtol=1e-12, atol=1e-12)   # ILU(0) of a block-diagonal matrix of dense blocks is its LU: M^-1 r solves the system


@pytest.mark.parametrize("name", ["strip3_reversed", "channel_periodic_sliding", "two_by_two_junction", "strip2_40x300"])
def test_bicgstab_with_ilu0_in_the_solver_slot(name):
    # the slot as the reference defines it: assembled system in, both components solved, values after fillXSpecific / fillYSpecific passed
    # separately.  With ILU(0) as right preconditioner: same solution as the exact solve (<= 1e-10 rms), several times fewer iterations
    # than with the diagonal alone.
    import scipy.sparse.linalg as spla

    om, s = _system(name)
    s.fill_x_specific()
    vx = s.lhs_values.copy()
    s.fill_y_specific()
    vy = s.lhs_values.copy()
    p, ci = s.lhs_p.copy(), s.lhs_i.copy()
    bx, by = s.rhs_x.copy(), s.rhs_y.copy()
    x0 = om.flat()
    two = not np.array_equal(vx, vy)
    import scipy.sparse as sp

    Ax = sp.csr_matrix((vx, ci, p), shape=(len(p) - 1,) * 2)
    Ay = sp.csr_matrix((vy, ci, p), shape=(len(p) - 1,) * 2)
    ref_x, ref_y = spla.splu(Ax.tocsc()).solve(bx), spla.splu(Ay.tocsc()).solve(by)
    xd, yd, rcd, std = csr_solve(p, ci, vx, bx, by, x0[:, 0], x0[:, 1], Ay=vy if two else None, rtol=1e-13, max_inner=20000)
    xi, yi, rci, sti = csr_solve(p, ci, vx, bx, by, x0[:, 0], x0[:, 1], Ay=vy if two else None, rtol=1e-13, max_inner=2000, preconditioner=solver.Preconditioner.ilu0)
    assert rcd == 0 and rci == 0, (std, sti)
    rms = float(np.sqrt(np.mean(np.concatenate([xi - ref_x, yi - ref_y]) ** 2)))
    assert rms <= 1e-10, rms
    assert sti["inner_iterations"] * 3 <= std["inner_iterations"], (sti["inner_iterations"], std["inner_iterations"])


def test_the_matrix_free_path_refuses_ilu0():
    mesh = configs.single_block(17, 21)
    with pytest.raises(_capi.TmError) as e:
        smooth.mesh(mesh, 1, solver.Option.hip(preconditioner=solver.Preconditioner.ilu0))
    assert e.value.code == _capi.TM_E_UNSUPPORTED and "seam 2" in str(e.value)


@pytest.mark.parametrize("name", ["strip3_reversed", "channel_periodic_sliding", "two_by_two_junction"])
@pytest.mark.parametrize("inner,precond", [(solver.Inner.gmres, solver.Preconditioner.diagonal), (solver.Inner.gmres, solver.Preconditioner.ilu0),
                                           (solver.Inner.bicgstab, solver.Preconditioner.diagonal), (solver.Inner.bicgstab, solver.Preconditioner.ilu0)])
def test_the_four_solver_preconditioner_combinations_of_the_reference_in_the_slot(name, inner, precond):
    # solver.zig:18-27: gmres / bicgstab, each with preconditioner diagonal / ilu0 -- all four on the device in the linear-solver slot, on the
    # caller's assembled system, both components; every one lands on the exact solve's solution (<= 1e-10 rms)
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    om, s = _system(name)
    s.fill_x_specific()
    vx = s.lhs_values.copy()
    s.fill_y_specific()
    vy = s.lhs_values.copy()
    p, ci = s.lhs_p.copy(), s.lhs_i.copy()
    bx, by = s.rhs_x.copy(), s.rhs_y.copy()
    n = len(p) - 1
    ref_x = spla.splu(sp.csr_matrix((vx, ci, p), shape=(n, n)).tocsc()).solve(bx)
    ref_y = spla.splu(sp.csr_matrix((vy, ci, p), shape=(n, n)).tocsc()).solve(by)
    x0 = om.flat()
    two = not np.array_equal(vx, vy)
    x, y, rc, st = csr_solve(p, ci, vx, bx, by, x0[:, 0], x0[:, 1], Ay=vy if two else None, rtol=1e-13, max_inner=30000, inner=inner, preconditioner=precond)
    assert rc == 0, (inner.name, precond.name, st)
    rms = float(np.sqrt(np.mean(np.concatenate([x - ref_x, y - ref_y]) ** 2)))
    assert rms <= 1e-10, (inner.name, precond.name, rms, st["inner_iterations"])
