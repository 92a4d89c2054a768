"""Seam 2 (reference src/core/smoothing/solver.zig:40-93): tm_csr_solve -- a caller-assembled CSR system solved on the MI355X.

The matrix here is built by the ORACLE's faithful restatement of the reference's assembly (smooth.zig:421-1165) or is the
reference's own known-answer system (umfpack.zig:71-97), never by the device kernels: an operator check that does not pass
through the matrix-free path or its mirror."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat, oracle_tfi
from tests.meshes import TOPOLOGIES
from turbomesh_amd import _capi
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu
_ip = C.POINTER(C.c_int32)


def csr_solve(Ap, Ai, Ax, bx, by, x0=None, y0=None, Ay=None, **opt):
    Ap = np.ascontiguousarray(Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Ax = np.ascontiguousarray(Ax, dtype=np.float64)
    n = len(Ap) - 1
    bx, by = np.ascontiguousarray(bx, dtype=np.float64), np.ascontiguousarray(by, dtype=np.float64)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    y = np.zeros(n) if y0 is None else np.array(y0, dtype=np.float64)
    o = solver.Option.hip(**opt).c_struct()
    st = _capi.tm_stats()
    ay = None if Ay is None else np.ascontiguousarray(Ay, dtype=np.float64)
    rc = _capi.check(_capi.lib().tm_csr_solve(n, Ap.ctypes.data_as(_ip), Ai.ctypes.data_as(_ip), _capi.f64ptr(Ax), None if ay is None else _capi.f64ptr(ay),
                                              _capi.f64ptr(bx), _capi.f64ptr(by), _capi.f64ptr(x), _capi.f64ptr(y), C.byref(o), C.byref(st)))
    return x, y, rc, st.as_dict()


def test_umfpack_5x5_known_answer():
    # umfpack.zig:71-97 (CSC): A x = b with x = (1,2,3,4,5); the y-system gets 2 b -> 2 x.  Two rows have no diagonal entry
    # (D^-1 := 1 there, BiCGStab.zig:169-173)
    import scipy.sparse as sp

    A = sp.csc_matrix(([2.0, 3.0, 3.0, -1.0, 4.0, 4.0, -3.0, 1.0, 2.0, 2.0, 6.0, 1.0], [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4], [0, 2, 5, 9, 10, 12]),
                      shape=(5, 5)).tocsr()
    A.sort_indices()
    b = np.array([8.0, 45.0, -3.0, 3.0, 19.0])
    x, y, rc, st = csr_solve(A.indptr, A.indices, A.data, b, 2 * b, rtol=1e-14, max_inner=200, check_every=1)
    assert rc == 0, st
    assert np.abs(x - np.arange(1, 6)).max() < 1e-9 and np.abs(y - 2 * np.arange(1, 6)).max() < 1e-9


def test_petsc_diagonal_known_answer():
    # petsc.zig:110-178: diag(1, 2) x = (1, 4) -> x = (1, 2), the reference's other sparse-solver known answer (1e-15 there)
    x, y, rc, st = csr_solve([0, 1, 2], [0, 1], [1.0, 2.0], [1.0, 4.0], [2.0, 8.0], rtol=1e-15, max_inner=50, check_every=1)
    assert rc == 0, st
    assert np.abs(x - [1.0, 2.0]).max() <= 1e-15 and np.abs(y - [2.0, 4.0]).max() <= 1e-15


def test_spd_laplacian_closed_form_and_warm_start():
    # tridiag(-1,2,-1) x = e_1 -> x_i = (n - i)/(n + 1)
    import scipy.sparse as sp

    n = 60
    A = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    b = np.zeros(n)
    b[0] = 1.0
    exact = (n - np.arange(n)) / (n + 1.0)
    x, y, rc, st = csr_solve(A.indptr, A.indices, A.data, b, -b, rtol=1e-13, check_every=1)
    assert rc == 0 and np.abs(x - exact).max() < 1e-10 and np.abs(y + exact).max() < 1e-10
    assert st["inner_iterations"] > 0
    x2, y2, rc2, st2 = csr_solve(A.indptr, A.indices, A.data, b, -b, x0=x, y0=y, rtol=1e-10, check_every=1)   # x_new / y_new carry the guess in
    assert rc2 == 0 and st2["inner_iterations"] == 0 and np.array_equal(x2, x)


def test_invalid_matrix_and_foreign_tag_are_refused():
    with pytest.raises(_capi.TmError) as e:
        csr_solve([0, 1, 2], [0, 5], [1.0, 1.0], [1, 1], [1, 1])          # column index out of range (error.InvalidMatrix, umfpack.zig:46)
    assert e.value.code == _capi.TM_E_ARG
    o = solver.Option(tag=solver.Tag.umfpack).c_struct()
    a = np.array([1.0]), np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32)
    x = np.zeros(1)
    rc = _capi.lib().tm_csr_solve(1, a[1].ctypes.data_as(_ip), a[2].ctypes.data_as(_ip), _capi.f64ptr(a[0]), None, _capi.f64ptr(a[0]), _capi.f64ptr(a[0]),
                                  _capi.f64ptr(x), _capi.f64ptr(x.copy()), C.byref(o), None)
    assert rc == _capi.TM_E_UNSUPPORTED   # error.ExternalSolverNotEnabled (solver.zig:48)


class _System:
    """The reference's RowCompressedMatrixSystem2d as the backend sees it (smooth.zig:277-307), backed by the oracle's assembly."""

    def __init__(self, s):
        self._s = s
        self.lhs_p, self.lhs_i = s.lhs_p, s.lhs_i
        self.rhs_x, self.rhs_y, self.x_new, self.y_new = s.rhs_x, s.rhs_y, s.x_new, s.y_new

    @property
    def lhs_values(self):
        return self._s.lhs_values

    def fillXSpecific(self):
        self._s.fill_x_specific()

    def fillYSpecific(self):
        self._s.fill_y_specific()


@pytest.mark.parametrize("name", ["two_by_two_junction", "channel_periodic_sliding", "strip3_reversed", "plate_le"])
def test_reference_assembled_system_solved_on_the_gpu_equals_exact_picard(name):
    # smooth.mesh with the `hip` backend in the solver slot: oracle-assembled CSR (the reference's rows, column order and all),
    # GPU-solved, committed by the oracle's copy-back -- every Picard iterate equals the exact-solve iterate.  The sliding
    # channel exercises fillXSpecific / fillYSpecific (the two systems differ).
    mesh = TOPOLOGIES[name](oracle_tfi)
    control = ("white", 0.02, 0.5 * np.pi) if name == "plate_le" else None
    ref = OracleMesh(mesh)
    hist, iterates = oracle.picard_exact(ref, 3, control=control, keep_iterates=True)
    om = OracleMesh(mesh)
    s = oracle.System(om, control)
    s.seed_initial_guess()
    sol = solver.Solver.init(solver.Option.hip(rtol=1e-13, max_inner=5000), _System(s))
    for n in range(3):
        s.fill(n)
        assert sol.solve(), sol.stats
        res = s.commit()[0]
        got = om.flat()
        want = np.concatenate([b.reshape(-1, 2) for b in iterates[n]])
        rms = float(np.sqrt(np.mean((got - want) ** 2)))
        assert rms <= 1e-10, (name, n, rms)
        assert res == pytest.approx(hist[n], rel=1e-5, abs=1e-30)
    s.close()


def test_t106_oracle_assembled_system_vs_the_matrix_free_path():
    # the oracle-assembled T106 system (8 blocks, 21 connections, 12 junction points, 3 periodic interfaces, 2 sliding boundaries)
    # solved through seam 2 against (a) the exact solve and (b) the matrix-free device path (seam 1) on the same mesh
    from tests.test_o4h import load

    inp, mesh = load("T106", oracle_tfi)
    ref = OracleMesh(mesh)
    hist, _ = oracle.picard_exact(ref, 1)
    om = OracleMesh(mesh)
    s = oracle.System(om)
    s.seed_initial_guess()
    s.fill(0)
    sol = solver.Solver.init(solver.Option.hip(rtol=1e-13, max_inner=20000), _System(s))
    assert sol.solve(), sol.stats
    s.commit()
    rms = float(np.sqrt(np.mean((om.flat() - ref.flat()) ** 2)))
    assert rms <= 1e-10, rms
    st = smooth.mesh(mesh, 1, solver.Option.hip(rtol=1e-13, max_inner=20000))
    assert st["not_converged"] == 0
    assert float(np.sqrt(np.mean((mesh_flat(mesh) - om.flat()) ** 2))) <= 1e-10
    s.close()
