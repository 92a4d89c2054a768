"""How far is "the exact Picard iterate" from itself?  (VERDICT r2, weak #2.)

The parity bar of north_star -- coordinates within 1e-10 RMS of the CPU reference -- compares the GPU with the reference run with a
solver that solves its systems exactly (its UMFPACK backend, umfpack.zig:18-24), restated as oracle-assembled CSR + sparse LU.  Two
sparse LUs with different elimination orders (SuperLU with COLAMD and with MMD(A'+A) column orderings) are two such exact solvers;
their mutual distance over the ten iterations of the example JSONs is the honest floor for any tolerance asserted against "the exact
iterate".  Measured here (container, scipy 1.15 / SuperLU):

    LS89 (white from its JSON): 1.0e-15 ... 1.5e-15 RMS at every one of the ten iterates
    T106 (white from its JSON): 1.7e-14 7.5e-15 6.9e-15 3.7e-14 2.3e-13 1.6e-13 2.4e-12 4.1e-12 1.3e-10 1.2e-10

T106's White update (wall_control_function.zig:282-320) still moves the mesh by 1e-4 RMS per iteration at iterations 7-10 and
amplifies rounding-level differences a hundredfold per two iterations: NO two solvers agree to 1e-10 there, the oracle with itself
included.  tests/test_gpu_benchsize.py::test_o4h_json_as_written_all_ten_iterations_with_default_options therefore holds the GPU to
max(1e-10, 3 x the running maximum of this distance) per iterate, computed in the same run."""
import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from tests.test_o4h import load


def self_distance(mesh, control, iterations):
    """(rms between the two exact solvers per iterate, iterates of the COLAMD run)"""
    a = oracle.picard_exact(OracleMesh(mesh), iterations, control=control, keep_iterates=True)[1]
    b = oracle.picard_exact(OracleMesh(mesh), iterations, control=control, keep_iterates=True, permc_spec="MMD_AT_PLUS_A")[1]
    fa = [np.concatenate([x.reshape(-1, 2) for x in it], axis=0) for it in a]
    fb = [np.concatenate([x.reshape(-1, 2) for x in it], axis=0) for it in b]
    return [float(np.sqrt(np.mean((p - q) ** 2))) for p, q in zip(fa, fb)], a


@pytest.mark.parametrize("name", ["T106", "LS89"])
def test_two_exact_solvers_over_the_jsons_ten_iterations(name):
    inp, mesh = load(name, oracle_tfi)
    w = inp.wall_control_function.white
    d, _ = self_distance(mesh, ("white", w.ds_target, w.theta_target), inp.iterations)
    print(f"[self-distance] {name}: " + " ".join(f"{x:.1e}" for x in d))
    assert max(d[:4]) <= 1e-12, d                       # both meshes: the first iterates are reproducible to rounding
    if name == "LS89":
        assert max(d) <= 1e-13, d                        # LS89 stays there
    else:
        assert max(d[6:]) >= 10.0 * max(d[:4]), d        # T106 does not: the iteration itself amplifies rounding (see docstring)


def test_single_block_floor_small():
    # a perturbed 129^2 block, Laplace: splu (two orderings) and the oracle's banded LU (a third elimination order) agree to ~1e-15
    from turbomesh_amd import configs

    mesh = configs.single_block(129, 129, tfi=oracle_tfi, perturb=0.25)
    d, a = self_distance(mesh, None, 2)
    c = oracle.picard_direct(OracleMesh(mesh), 2, keep_iterates=True)[1]
    e = [float(np.sqrt(np.mean((x[0] - y[0]) ** 2))) for x, y in zip(a, c)]
    assert max(d) <= 1e-13 and max(e) <= 1e-13, (d, e)
