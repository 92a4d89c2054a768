"""GMRES(30) on the device (TM_INNER_GMRES, csrc/tm_gmres.hip) -- the reference's other Krylov solver (GMRES.zig:300-423: left-preconditioned
restarted GMRES, modified Gram-Schmidt, Givens rotations :510-524) with the diagonal preconditioner, the solver both example inputs name.

Parity statement, as for BiCGStab: a Krylov method changes the route, not the destination -- every Picard iterate equals the exact-solve
oracle's (the reference with its UMFPACK backend, umfpack.zig:18-24) to <= 1e-10 RMS, on the ten test topologies, with the White control
function and on T106 / LS89.  The recurrence itself is checked against the faithful CPU restatement of GMRES.zig (oracle/orc_solvers.cpp)
solving the same frozen system with the same tolerance: same iteration count (give or take rounding at the stop test), same solution."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, mesh_flat
from tests.meshes import TOPOLOGIES
from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

pytestmark = pytest.mark.gpu
TOL_RMS = 1e-10


def _rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("name", list(TOPOLOGIES))
def test_picard_iterates_with_gmres_match_exact_oracle(name):
    mesh = TOPOLOGIES[name]()
    om = OracleMesh(mesh)
    hist, iterates = oracle.picard_exact(om, 3, keep_iterates=True)
    # default options: inner rtol 1e-14 at these sizes (the distance from the exact iterate is conditioning x tolerance; the minimal-residual
    # iterate of GMRES leaves its error in the smooth modes: strip2_40x300 lands 1.5e-10 from the exact iterate at 1e-13)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.gmres, max_inner=40000)) as sm:
        assert sm.inner == solver.Inner.gmres
        for k in range(3):
            st = sm.iterate(1)
            assert st["not_converged"] == 0, (name, k, st)
            sm.download()
            ref = np.concatenate([b.reshape(-1, 2) for b in iterates[k]])
            assert _rms(mesh_flat(mesh), ref) <= TOL_RMS, (name, k, _rms(mesh_flat(mesh), ref))
            assert st["last_residual"] == pytest.approx(hist[k], rel=1e-6, abs=1e-30)


def test_gmres_and_bicgstab_reach_the_same_iterates_and_fixed_point():
    build = lambda: configs.single_block(129, 97, perturb=0.25)
    a, b = build(), build()
    sa = smooth.mesh(a, 3, solver.Option.hip(inner=solver.Inner.gmres))
    sb = smooth.mesh(b, 3, solver.Option.hip(inner=solver.Inner.bicgstab))
    assert sa["not_converged"] == 0 and sb["not_converged"] == 0
    assert _rms(mesh_flat(a), mesh_flat(b)) <= 1e-11
    # GMRES(30) with the diagonal alone needs more operator applications than BiCGStab on this system -- it is here for parity of the
    # feature, not for speed (one application per inner iteration against two)
    assert sa["inner_iterations"] > 0 and sa["operator_sweeps"] >= sa["inner_iterations"]
    c, d = build(), build()
    with smooth.Smoother(c, solver.Option.hip(inner=solver.Inner.gmres)) as sm:
        reached, st = sm.iterate_until_update(1e-11, 60)
        sm.download()
    assert reached
    with smooth.Smoother(d, solver.Option.hip()) as sm:
        reached, _ = sm.iterate_until_update(1e-11, 60)
        sm.download()
    assert reached and _rms(mesh_flat(c), mesh_flat(d)) <= 1e-10


def test_device_recurrence_against_the_faithful_cpu_gmres_on_one_frozen_system():
    # one frozen system (the first Picard solve of a perturbed block), tolerance on the scaled residual: the oracle's restatement of
    # GMRES.zig with the diagonal preconditioner solves D^-1 A x = D^-1 b when the system is row-equilibrated first -- done here through
    # scipy on the oracle-assembled CSR -- and must take the same number of Arnoldi steps as the device, give or take the rounding at
    # the stop test, and land on the same solution
    import scipy.sparse as sp

    mesh = configs.single_block(41, 37, perturb=0.25)
    om = OracleMesh(mesh)
    s = oracle.System(om)
    s.fill(0)
    A = s.csr()
    d = A.diagonal()
    As = sp.diags(1.0 / d) @ A
    rtol = 1e-12
    x_dev = configs.single_block(41, 37, perturb=0.25)
    with smooth.Smoother(x_dev, solver.Option.hip(inner=solver.Inner.gmres, rtol=rtol, check_every=1)) as sm:
        st = sm.iterate(1)
        sm.download()
    for c, b in enumerate((s.rhs_x, s.rhs_y)):
        bs = b / d
        x0 = om.blocks[0].reshape(-1, 2)[:, c].copy()
        # textbook restarted GMRES(30) with MGS on the equilibrated system, stop at |g_{j+1}| <= rtol ||bs|| (numpy, a few lines: the same
        # algorithm as GMRES.zig:300-423 with M = I)
        x, its = _gmres30(As, bs, x0, rtol * np.linalg.norm(bs))
        got = x_dev.blocks[0].points.data.reshape(-1, 2)[:, c]
        assert np.abs(got - x).max() <= 1e-9 * max(1.0, np.abs(x).max()), c
        assert np.linalg.norm(bs - As @ got) <= 10 * rtol * np.linalg.norm(bs)
    assert st["not_converged"] == 0 and st["inner_iterations"] > 30     # at least one restart happened


def _gmres30(A, b, x, tol, m=30, cap=20000):
    its = 0
    while its < cap:
        r = b - A @ x
        beta = np.linalg.norm(r)
        if beta <= tol:
            return x, its
        V = [r / beta]
        H = np.zeros((m + 1, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        g[0] = beta
        used, conv = 0, False
        for j in range(m):
            z = A @ V[j]
            for i in range(j + 1):
                H[i, j] = z @ V[i]
                z = z - H[i, j] * V[i]
            H[j + 1, j] = np.linalg.norm(z)
            V.append(z / H[j + 1, j] if H[j + 1, j] > 1e-30 else z)
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            a, bb = H[j, j], H[j + 1, j]
            if bb == 0.0:
                c_, s_, r_ = 1.0, 0.0, a
            elif abs(bb) > abs(a):
                t = a / bb
                s_ = 1.0 / np.sqrt(1.0 + t * t)
                c_, r_ = s_ * t, bb / s_
            else:
                t = bb / a
                c_ = 1.0 / np.sqrt(1.0 + t * t)
                s_, r_ = c_ * t, a / c_
            cs[j], sn[j] = c_, s_
            H[j, j], H[j + 1, j] = r_, 0.0
            g[j], g[j + 1] = c_ * g[j] + s_ * g[j + 1], -s_ * g[j] + c_ * g[j + 1]
            its += 1
            used = j + 1
            if abs(g[j + 1]) <= tol:
                conv = True
                break
        y = np.linalg.solve(np.triu(H[:used, :used]), g[:used])
        for i in range(used):
            x = x + y[i] * V[i]
        if conv:
            return x, its
    return x, its


def test_gmres_with_the_white_control_function_on_the_plate():
    mesh = TOPOLOGIES["plate_le"]()
    om = OracleMesh(mesh)
    _, iterates = oracle.picard_exact(om, 3, keep_iterates=True, control=("white", 0.02, 0.5 * np.pi))
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.gmres, max_inner=40000), wcf.Algorithm(wcf.White(0.02))) as sm:
        for k in range(3):
            st = sm.iterate(1)
            assert st["not_converged"] == 0
            sm.download()
            ref = np.concatenate([b.reshape(-1, 2) for b in iterates[k]])
            assert _rms(mesh_flat(mesh), ref) <= TOL_RMS, (k, _rms(mesh_flat(mesh), ref))


def test_gmres_iteration_cap_is_a_warning_not_an_error():
    mesh = configs.single_block(65, 65, perturb=0.25)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.gmres, max_inner=7)) as sm:
        st = sm.iterate(1)               # GMRES.zig:422: log.warn, the iterate is kept
        assert st["not_converged"] == 1 and st["inner_iterations"] == 7
        sm.download()
    assert np.isfinite(mesh.blocks[0].points.data).all()
