"""The C++ host mirror (turbomesh_amd/host/turbomesh.hpp) + harness (the stand-in for the Zig caller,
reference src/gui/main.zig:30-56): same flow, same log lines, results checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import configs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "turbomesh_amd", "tm_harness")


def test_harness_built_and_fails_loudly_without_arguments():
    assert os.path.exists(HARNESS), "build with __graft_entry__.build()"
    r = subprocess.run([HARNESS], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,builder", [
    (["strip", "3", "17", "24", "3", "bicgstab"], lambda: configs.strip(3, 17, 24, tfi=oracle_tfi)),
    (["single", "33", "41", "3", "bicgstab"], lambda: configs.single_block(33, 41, tfi=oracle_tfi)),
    (["strip", "3", "33", "40", "3", "mg"], lambda: configs.strip(3, 33, 40, tfi=oracle_tfi)),   # multigrid-preconditioned inner solve
    (["strip", "3", "17", "24", "3", "gmres"], lambda: configs.strip(3, 17, 24, tfi=oracle_tfi)),  # GMRES(30) on the device, default tolerance
    (["single", "33", "41", "3", "auto"], lambda: configs.single_block(33, 41, tfi=oracle_tfi)),    # the size-aware choice (here: bicgstab)
])
def test_harness_matches_oracle(tmp_path, args, builder):
    dump = str(tmp_path / "dump.bin")
    r = subprocess.run([HARNESS] + args + [dump], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "info(smoothing): \tresidual:" in r.stdout and "elapsed time for smoothing" in r.stdout   # smooth.zig:137, 159
    got = np.fromfile(dump, dtype=np.float64).reshape(-1, 2)
    om = OracleMesh(builder())
    hist, _ = oracle.picard_exact(om, 3)
    assert float(np.sqrt(np.mean((got - om.flat()) ** 2))) <= 1e-10
    # the reference's two lines per outer iteration (smooth.zig:105, 136-137): every iteration announced, every residual logged
    lines = r.stdout.splitlines()
    assert [l for l in lines if "iteration:" in l] == [f"info(smoothing): iteration: {n}" for n in range(3)]
    logged = [float(l.split("residual:")[1]) for l in lines if "residual:" in l]
    assert len(logged) == 3 and logged == pytest.approx(list(hist), rel=1e-6)


@pytest.mark.gpu
def test_harness_config4_shape_runs():
    # SURVEY 8d config 4 topology at a size that runs in a second: 8 blocks, 7 interfaces, relax sweeps
    r = subprocess.run([HARNESS, "strip", "8", "256", "256", "50", "relax"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "operator_sweeps 50" in r.stdout


@pytest.mark.gpu
def test_harness_handle_iterate_until_and_plot3d(tmp_path):
    # the C++ mirror's device-resident handle: iterate to a residual, write PLOT3D planes transposed on the device
    from turbomesh_amd import output

    dump, p3d = str(tmp_path / "dump.bin"), str(tmp_path / "mesh.xyz")
    r = subprocess.run([HARNESS, "strip", "2", "65", "70", "40", "mg", "until", "1e-11", "write", p3d, dump], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "reached 1" in r.stdout
    got = np.fromfile(dump, dtype=np.float64).reshape(2, 65, 70, 2)
    blocks = output.read_plot3d(p3d)
    assert [(b[0], b[1]) for b in blocks] == [(65, 70), (65, 70)]
    for k, (ni, nj, x, y) in enumerate(blocks):
        assert np.array_equal(x, got[k, :, :, 0]) and np.array_equal(y, got[k, :, :, 1])
    # a format that is not built in answers like the reference without the cgns library
    r = subprocess.run([HARNESS, "single", "9", "9", "1", "bicgstab", "write", str(tmp_path / "m.cgns")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "OutputFormatNotEnabled" in r.stderr


@pytest.mark.gpu
def test_harness_csr_slot_known_answer():
    # seam 2 (solver.zig:40-93) from the compiled caller: umfpack.zig:71-97's system through tm_csr_solve
    r = subprocess.run([HARNESS, "csr"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "csr kat: rc 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("solver_name", ["relax", "bicgstab"])
def test_harness_drives_the_rccl_transport(tmp_path, solver_name):
    # a compiled caller through tm_rccl_unique_id / tm_rccl_comm_create / tm_rccl_hooks (the one-GPU box allows a one-rank
    # communicator: rendezvous by file, hooks filled by the library, handle created with them) == the same job without hooks.
    # relax: bit for bit; bicgstab: the hooked handle runs the launch-per-step recurrence, the plain one the two-kernel one.
    import torch

    env = dict(os.environ, TM_LIBRCCL=os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    its = "7" if solver_name == "relax" else "2"
    d1, d2 = str(tmp_path / "ranks.bin"), str(tmp_path / "plain.bin")
    r = subprocess.run([HARNESS, "ranks", "1", "0", str(tmp_path / "id"), "2", "40", "130", its, solver_name, d1], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr
    assert "rank 0 of 1" in r.stdout
    r = subprocess.run([HARNESS, "strip", "2", "40", "130", its, solver_name, d2], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    a, b = np.fromfile(d1, dtype=np.float64), np.fromfile(d2, dtype=np.float64)
    assert a.shape == b.shape == (2 * 40 * 130 * 2,)
    if solver_name == "relax":
        assert np.array_equal(a, b)
    else:
        assert float(np.sqrt(np.mean((a - b) ** 2))) <= 2e-10
    # rank outside the job / fewer blocks than ranks: refused before anything collective happens
    r = subprocess.run([HARNESS, "ranks", "2", "2", str(tmp_path / "id2"), "2", "40", "130", "1"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 1 and "error(" in r.stderr
