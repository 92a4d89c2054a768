"""N3 structured output: the device transpose (K8) against a numpy restatement of cgns.zig:75-104, and the PLOT3D files."""
import ctypes as C
import os

import numpy as np
import pytest

from turbomesh_amd import _capi, configs, output
from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf


def _planes_oracle(points):
    """cgns.zig:75-104 restated: for j: for i: buffer[idx++] = block(i,j).x  (test infrastructure)"""
    return points[:, :, 0].T.ravel().copy(), points[:, :, 1].T.ravel().copy()


def test_plot3d_files_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    blocks = [rng.standard_normal((5, 7, 2)), rng.standard_normal((33, 4, 2))]
    sizes = [b.shape[:2] for b in blocks]
    fn = os.path.join(tmp_path, "m.xyz")
    output.write_plot3d(fn, sizes, [_planes_oracle(b) for b in blocks])
    back = output.read_plot3d(fn)
    for b, (ni, nj, x, y) in zip(blocks, back):
        assert (ni, nj) == b.shape[:2] and np.array_equal(x, b[:, :, 0]) and np.array_equal(y, b[:, :, 1])
    assert os.path.getsize(fn) == 4 + 8 * 2 + 16 * (5 * 7 + 33 * 4)
    with pytest.raises(output.OutputFormatNotEnabled):   # discrete.zig:215: a build without the cgns library
        output._format_of("mesh.cgns")


@pytest.mark.gpu
@pytest.mark.parametrize("ni,nj", [(1, 1), (2, 3), (31, 33), (32, 32), (33, 65), (100, 257), (1000, 37)])
def test_export_planes_bit_exact(ni, nj):
    rng = np.random.default_rng(ni * 1000 + nj)
    pts = rng.standard_normal((ni, nj, 2))
    x, y = output.block_planes(pts)
    ex, ey = _planes_oracle(pts)
    assert np.array_equal(x, ex) and np.array_equal(y, ey)


@pytest.mark.gpu
def test_mesh_and_smoother_write(tmp_path):
    mesh = configs.plate(15, 9)
    fn = os.path.join(tmp_path, "plate.p3d")
    mesh.write(fn)
    for blk, (ni, nj, x, y) in zip(mesh.blocks, output.read_plot3d(fn)):
        assert np.array_equal(x, blk.points.data[:, :, 0]) and np.array_equal(y, blk.points.data[:, :, 1])
    with pytest.raises(output.OutputFormatNotEnabled):
        mesh.write(os.path.join(tmp_path, "plate.cgns"))
    # handle: resident coordinates after two iterations + the White control function as P, Q planes
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-12), wcf.Algorithm(wcf.White(0.02, 0.5 * np.pi))) as sm:
        sm.iterate(2)
        fs = os.path.join(tmp_path, "smooth.xyz")
        sm.write(fs)
        sm.download()
        pq = np.empty((sm.dof, 2))
        _capi.check(_capi.lib().tm_smoother_control_function(sm._h, pq.ctypes.data_as(C.POINTER(C.c_double))))
    off = 0
    back = output.read_plot3d(fs)
    with open(os.path.join(tmp_path, "smooth.f"), "rb") as f:
        nb = int(np.fromfile(f, dtype="<i4", count=1)[0])
        hdr = np.fromfile(f, dtype="<i4", count=3 * nb).reshape(nb, 3)
        assert nb == len(mesh.blocks) and np.all(hdr[:, 2] == 2)
        for blk, (ni, nj, x, y) in zip(mesh.blocks, back):
            assert np.array_equal(x, blk.points.data[:, :, 0]) and np.array_equal(y, blk.points.data[:, :, 1])
            P = np.fromfile(f, dtype="<f8", count=ni * nj).reshape(nj, ni).T
            Q = np.fromfile(f, dtype="<f8", count=ni * nj).reshape(nj, ni).T
            ref = pq[off:off + ni * nj].reshape(ni, nj, 2)
            assert np.array_equal(P, ref[:, :, 0]) and np.array_equal(Q, ref[:, :, 1])
            off += ni * nj
    assert np.abs(pq).max() > 0   # the wall control function is active


@pytest.mark.gpu
@pytest.mark.parametrize("single", [True, False], ids=["relax", "relax2"])
@pytest.mark.parametrize("build", [lambda: configs.single_block(70, 131, perturb=0.2), lambda: configs.strip(3, 33, 70, reverse_odd=True)], ids=["single", "strip3"])
def test_write_between_relax_sweeps_leaves_the_handle_untouched(tmp_path, single, build):
    # a relax handle keeps the `fixed` boundary coordinates on the perimeter of EVERY field buffer between iterate() calls
    # (prefill_fixed); the export must not use any of them as scratch: iterate(3); write; iterate(3) == iterate(6), bit for bit
    opt = solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)
    a, b = build(), build()
    with smooth.Smoother(a, opt) as sm:
        sm.iterate(6)
        sm.download()
    with smooth.Smoother(b, opt) as sm:
        sm.iterate(3)
        sm.write(os.path.join(tmp_path, "mid.xyz"))
        sm.download()
        for blk, (ni, nj, x, y) in zip(b.blocks, output.read_plot3d(os.path.join(tmp_path, "mid.xyz"))):
            assert np.array_equal(x, blk.points.data[:, :, 0]) and np.array_equal(y, blk.points.data[:, :, 1])
        sm.iterate(3)
        sm.download()
    for p, q in zip(a.blocks, b.blocks):
        assert np.array_equal(p.points.data, q.points.data)
