"""CPU: pin the oracle (and the host mirror) on every known-answer vector the reference's own tests hold
next to the hot path, plus the analytic known answers of SURVEY.md 8c.

  - Edge.combine exact vectors                 reference src/core/discrete.zig:219-290
  - 5x5 sparse system, x = (1,2,3,4,5)         reference src/core/smoothing/umfpack.zig:71-97
  - commented 3x3 TFI case                     reference src/core/tfi.zig:230-260
  - Mat2d NaN initialisation                   reference src/core/types.zig:103-114
The reference pins neither TFI nor smoothing output ("parity unpinned")."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import clustering, configs
from turbomesh_amd.discrete import Edge, EdgeView, Mesh
from turbomesh_amd.geometry import Line
from turbomesh_amd.types import Mat2d, Vec2d


def _two_edges(make):
    e1 = make(3, (0.0, 0.0), (2.0, 0.0))
    e2 = make(3, (2.0, 0.0), (4.0, 0.0))
    return e1, e2


CASES = [   # (views as (edge index, start, end), expected points x, expected clustering) -- discrete.zig:227-289
    ([(0, 0, 2), (1, 0, 2)], [0, 1, 2, 3, 4], [0.0, 0.25, 0.5, 0.75, 1.0]),
    ([(0, 1, 2), (1, 0, 1)], [1, 2, 3], [0.0, 0.5, 1.0]),
    ([(1, 2, 0), (0, 2, 0)], [4, 3, 2, 1, 0], [0.0, 0.25, 0.5, 0.75, 1.0]),
    ([(1, 1, 0), (0, 2, 1)], [3, 2, 1], [0.0, 0.5, 1.0]),
]


@pytest.mark.parametrize("views,xs,cl", CASES)
def test_edge_combine_reference_vectors_oracle(views, xs, cl):
    def make(n, a, b):
        u = oracle.cluster_uniform(n)
        return oracle.line_interpolate(a, b, u), u

    edges = _two_edges(make)
    pts, u = oracle.edge_combine([(edges[k][0], edges[k][1], s, e) for k, s, e in views])
    assert np.array_equal(pts, np.array([[x, 0.0] for x in xs], float))   # expectEqualDeep: exact
    assert np.array_equal(u, np.array(cl))


@pytest.mark.parametrize("views,xs,cl", CASES)
def test_edge_combine_reference_vectors_host_mirror(views, xs, cl):
    def make(n, a, b):
        return Edge.init(n, Line(Vec2d(*a), Vec2d(*b)), clustering.Uniform())

    edges = _two_edges(make)
    out = Edge.combine([EdgeView(edges[k], s, e) for k, s, e in views])
    assert np.array_equal(out.points, np.array([[x, 0.0] for x in xs], float))
    assert np.array_equal(out.clustering, np.array(cl))


def test_umfpack_5x5_kat_all_oracle_solvers():
    # umfpack.zig:71-97 gives the matrix in CSC; as CSR it is the transpose, so the KAT is A^T... the reference's
    # smoothing call solves with UMFPACK_Aat for exactly that reason (umfpack.zig:54).  Build A explicitly.
    Ap = [0, 2, 5, 9, 10, 12]
    Ai = [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4]
    Ax = [2.0, 3.0, 3.0, -1.0, 4.0, 4.0, -3.0, 1.0, 2.0, 2.0, 6.0, 1.0]
    b = np.array([8.0, 45.0, -3.0, 3.0, 19.0])
    import scipy.sparse as sp

    A = sp.csc_matrix((Ax, Ai, Ap), shape=(5, 5)).tocsr()
    A.sort_indices()
    tol = np.finfo(float).eps * 10
    x, _, _ = oracle.csr_solve("direct", 5, A.indptr, A.indices, A.data, b)
    assert np.abs(x - np.arange(1, 6)).max() <= tol * 4
    for kind, pc in (("gmres", oracle.PRECOND_ILU0), ("gmres", oracle.PRECOND_DIAGONAL), ("bicgstab", oracle.PRECOND_ILU0)):
        x, it, ok = oracle.csr_solve(kind, 5, A.indptr, A.indices, A.data, b, precond=pc, rtol=1e-14, atol=1e-14)
        assert np.abs(x - np.arange(1, 6)).max() < 1e-9, (kind, pc, x)


def test_petsc_diagonal_kat_all_oracle_solvers():
    # petsc.zig:110-178: diag(1, 2) x = (1, 4) -> (1, 2) within 1e-15
    b = np.array([1.0, 4.0])
    x, _, _ = oracle.csr_solve("direct", 2, [0, 1, 2], [0, 1], [1.0, 2.0], b)
    assert np.abs(x - [1.0, 2.0]).max() <= 1e-15
    for kind, pc in (("gmres", oracle.PRECOND_ILU0), ("gmres", oracle.PRECOND_DIAGONAL), ("bicgstab", oracle.PRECOND_DIAGONAL), ("bicgstab", oracle.PRECOND_ILU0)):
        x, it, ok = oracle.csr_solve(kind, 2, [0, 1, 2], [0, 1], [1.0, 2.0], b, precond=pc, rtol=1e-15, atol=1e-15)
        assert np.abs(x - [1.0, 2.0]).max() <= 1e-15, (kind, pc, x)


def test_solvers_on_spd_laplacian_closed_form():
    # 1D Laplacian tridiag(-1,2,-1) x = e_1  ->  x_i = (n - i)/(n + 1), i = 0..n-1
    n = 40
    import scipy.sparse as sp

    A = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    b = np.zeros(n)
    b[0] = 1.0
    exact = (n - np.arange(n)) / (n + 1.0)
    for kind in ("bicgstab", "gmres", "direct"):
        x, it, ok = oracle.csr_solve(kind, n, A.indptr, A.indices, A.data, b, precond=oracle.PRECOND_ILU0, rtol=1e-13, atol=1e-13, max_iters=500)
        assert np.abs(x - exact).max() < 1e-10, kind


def test_tfi_commented_kat_and_edges_oracle():
    i_min = np.array([[0, 0], [0.5, 0], [1, 0]], float)
    i_max = np.array([[0, 2], [0.5, 2], [1, 2]], float)
    j_min = np.array([[0, 0], [0, 1], [0, 2]], float)
    j_max = np.array([[1, 0], [1, 1], [1, 2]], float)
    cl = np.array([0, 0.5, 1.0])
    out = oracle.tfi_block(i_min, i_max, j_min, j_max, cl, cl, cl, cl)
    assert out[1, 1, 0] == 0.5 and out[1, 1, 1] == 1.0
    assert np.array_equal(out[:, 0], i_min) and np.array_equal(out[:, -1], i_max)


def test_tfi_analytic_properties_oracle():
    # (i) straight-sided quadrilateral + uniform clustering = bilinear map; equal opposite clusterings -> u = s_i, v = t_j
    ni, nj = 9, 13
    s, t = oracle.cluster_uniform(ni), oracle.cluster_uniform(nj)
    P = {"00": np.array([0.0, 0.0]), "n0": np.array([2.0, 0.3]), "0m": np.array([-0.2, 1.5]), "nm": np.array([2.5, 2.0])}
    i_min = oracle.line_interpolate(P["00"], P["n0"], s)
    i_max = oracle.line_interpolate(P["0m"], P["nm"], s)
    j_min = oracle.line_interpolate(P["00"], P["0m"], t)
    j_max = oracle.line_interpolate(P["n0"], P["nm"], t)
    out = oracle.tfi_block(i_min, i_max, j_min, j_max, s, s, t, t)
    S, T = np.meshgrid(s, t, indexing="ij")
    bil = ((1 - S) * (1 - T))[..., None] * P["00"] + (S * (1 - T))[..., None] * P["n0"] + ((1 - S) * T)[..., None] * P["0m"] + (S * T)[..., None] * P["nm"]
    assert np.abs(out - bil).max() < 4e-15
    # clustered edges: the four block edges are reproduced up to rounding
    r = oracle.cluster_roberts(ni, 0.5, 1.1)
    h = oracle.cluster_tanh(nj, 0.02)
    assert r[0] == 0.0 and abs(r[-1] - 1.0) < 1e-15 and h[0] == 0.0 and h[-1] == 1.0 and np.all(np.diff(h) > 0)
    r[-1] = 1.0
    i_min, i_max = oracle.line_interpolate(P["00"], P["n0"], r), oracle.line_interpolate(P["0m"], P["nm"], r)
    j_min, j_max = oracle.line_interpolate(P["00"], P["0m"], h), oracle.line_interpolate(P["n0"], P["nm"], h)
    out = oracle.tfi_block(i_min, i_max, j_min, j_max, r, r, h, h)
    for got, want in ((out[:, 0], i_min), (out[:, -1], i_max), (out[0], j_min), (out[-1], j_max)):
        assert np.abs(got - want).max() < 4e-15


def test_mat2d_nan_init_and_index():
    m = Mat2d((20, 10))   # types.zig:103-114
    assert m.data.shape == (20, 10, 2) and np.isnan(m.data).all()
    assert m.index((3, 4)) == 4 + 10 * 3


def test_stencil_on_uniform_cartesian_grid():
    # spacing h (i) and k (j): centre -2(h^2+k^2), i-neighbours k^2, j-neighbours h^2, corners 0
    ni, nj, h, k = 7, 6, 0.25, 0.5
    I, J = np.meshgrid(np.arange(ni) * h, np.arange(nj) * k, indexing="ij")
    m = Mesh()
    m.addBlock("c", configs.block_from_array(np.stack([I, J], axis=2)))
    s = oracle.System(OracleMesh(m))
    s.fill(0)
    A = s.csr()
    row = 2 * nj + 3   # node (2,3)
    r = A.getrow(row).toarray().ravel()
    assert r[row] == -2 * (h * h + k * k)
    assert r[row - nj] == k * k and r[row + nj] == k * k and r[row - 1] == h * h and r[row + 1] == h * h
    for c in (row - nj - 1, row - nj + 1, row + nj - 1, row + nj + 1):
        assert r[c] == 0.0


def test_affine_fixed_point_oracle():
    # SURVEY 8c (iii)
    ni, nj = 11, 9
    i, j = np.meshgrid(np.arange(ni) / (ni - 1), np.arange(nj) / (nj - 1), indexing="ij")
    grid = np.stack([1.0 + 2.0 * i + 0.3 * j, -0.5 + 0.4 * i + 1.5 * j], axis=2)
    m = Mesh()
    m.addBlock("a", configs.block_from_array(grid.copy()))
    om = OracleMesh(m)
    st = oracle.smooth_mesh(om, 1, solver=oracle.SOLVER_DIRECT)
    assert st.last_residual < 1e-50
    om.blocks[0][1:-1, 1:-1] += 0.01 * (np.random.default_rng(1).random((ni - 2, nj - 2, 2)) - 0.5)
    oracle.smooth_mesh(om, 30, solver=oracle.SOLVER_DIRECT)
    assert np.abs(om.blocks[0] - grid).max() < 1e-12


def test_constraint_rows_oracle():
    # SURVEY 8c (iv) on the periodic channel with a sliding inlet and on the 2x2 junction mesh
    m = configs.periodic_channel(13, 9, tfi=oracle_tfi)
    om = OracleMesh(m)
    before = om.blocks[0].copy()
    oracle.smooth_mesh(om, 3, solver=oracle.SOLVER_DIRECT)
    d = om.blocks[0]
    assert np.abs(d[:, -1] - (d[:, 0] + np.array([0.0, 0.8]))).max() < 1e-13
    assert np.array_equal(d[0, 1:-1, 0], before[0, 1:-1, 0]) and np.abs(d[0, 1:-1, 1] - d[1, 1:-1, 1]).max() < 1e-13
    assert np.abs(d[-1] - before[-1]).max() < 1e-15
    m = configs.two_by_two(8, 9, tfi=oracle_tfi)
    om = OracleMesh(m)
    oracle.smooth_mesh(om, 3, solver=oracle.SOLVER_DIRECT)
    b0, b1, b2, b3 = om.blocks
    centre = b0[-1, -1]
    assert np.abs(centre - (b0[-2, -2] + b1[1, -2] + b2[-2, 1] + b3[1, 1]) / 4.0).max() < 1e-13
    for other in (b1[0, -1], b2[-1, 0], b3[0, 0]):
        assert np.abs(other - centre).max() < 1e-14


def test_reference_faithful_solvers_reach_their_own_tolerance():
    # BiCGStab.zig / GMRES.zig with the reference's fixed rtol=1e-6, atol=1e-8: close to, not equal to, the exact iterate
    m = configs.single_block(17, 21, tfi=oracle_tfi)
    exact = OracleMesh(m)
    oracle.picard_exact(exact, 3)
    for solver_tag, pc in ((oracle.SOLVER_BICGSTAB, oracle.PRECOND_DIAGONAL), (oracle.SOLVER_BICGSTAB, oracle.PRECOND_ILU0),
                           (oracle.SOLVER_GMRES, oracle.PRECOND_DIAGONAL), (oracle.SOLVER_GMRES, oracle.PRECOND_ILU0)):
        om = OracleMesh(m)
        st = oracle.smooth_mesh(om, 3, solver=solver_tag, preconditioner=pc)
        assert st.not_converged == 0 and st.inner_iterations > 0
        assert 0 < np.abs(om.flat() - exact.flat()).max() < 5e-3   # loose, unscaled reference tolerance
    om = OracleMesh(m)
    oracle.smooth_mesh(om, 3, solver=oracle.SOLVER_DIRECT)
    assert np.abs(om.flat() - exact.flat()).max() < 1e-13   # banded LU == scipy splu: the exact Picard iterate


def test_threaded_relax_sweeps_equal_single_thread():
    # bench.py's cpu_baseline reports the mirror sweep on 1 thread (faithful) and on all threads: same arithmetic, same bits
    i, j = np.meshgrid(np.linspace(0, 1, 70), np.linspace(0, 1, 131), indexing="ij")
    xy = np.stack([i + 0.01 * np.sin(7 * j), j + 0.01 * np.cos(5 * i)], axis=2).copy()
    a, b = xy.copy(), xy.copy()
    oracle.time_relax_sweeps(a, 5)
    oracle.time_relax_sweeps_mt(b, 5, 6)
    assert np.array_equal(a, b) and not np.array_equal(a, xy)


def test_mirror_jacobi_form_against_the_reference_order_csr():
    # The device's relaxation sweep with omega = 1 takes the textbook Jacobi form x_new = (b - sum_{k != i} a_ik x_k) / a_ii evaluated
    # from factored metric terms (oracle/orc_mirror.cpp MIRROR_RELAX, tm_kernels.hip winslow_row).  Tie it to the reference's own
    # arithmetic: the same step from the faithfully assembled CSR (coefficients smooth.zig:171-216, rows smooth.zig:923-992), term by
    # term in CSR order -- equal to rounding (a few eps of the neighbourhood's magnitude), on a rough field and on a stretched one.
    import scipy.sparse as sp

    for ni, nj, perturb in ((23, 31, 0.25), (9, 140, 0.1)):
        mesh = configs.single_block(ni, nj, tfi=oracle_tfi, perturb=perturb)
        om = OracleMesh(mesh)
        s = oracle.System(om)
        s.fill(0)
        A = s.csr()
        x = om.blocks[0].reshape(-1, 2).copy()
        d = A.diagonal()
        off = A - sp.diags(d)
        b = np.stack([s.rhs_x, s.rhs_y], axis=1)
        jac = (b - np.stack([off @ x[:, 0], off @ x[:, 1]], axis=1)) / d[:, None]
        mir = oracle.mirror_apply_block(oracle.MIRROR_RELAX, om.blocks[0], om.blocks[0], omega=1.0, out=om.blocks[0].copy()).reshape(-1, 2)
        interior = np.zeros((ni, nj), dtype=bool)
        interior[1:-1, 1:-1] = True
        m = interior.reshape(-1)
        scale = np.abs(x).max()
        assert np.abs(mir[m] - jac[m]).max() <= 32 * np.finfo(float).eps * scale, np.abs(mir[m] - jac[m]).max()
        assert np.abs(mir[m] - x[m]).max() > 1e-6           # the step does move the nodes
