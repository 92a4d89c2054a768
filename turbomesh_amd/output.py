"""Structured output of a mesh -- the step right after the hot path.

Mirrors the reference's writer boundary: `Mesh.write(filename)` (discrete.zig:197-216) and the smoother's
`system.write` with the control function (smooth.zig:396-414) hand the writer ONE PLANE PER COORDINATE with i fastest
(cgns.zig:75-104: plane[j*ni + i] = block(i,j)), optionally planes "P" and "Q" (cgns.zig:106-154).  That
de-interleaving transpose runs on the device (`tm_export_soa` / `tm_smoother_export_soa`, kernel K8).

File formats.  The reference writes CGNS through the system cgns library, which this image does not have: like a
reference build without -Duse-cgns, `.cgns` raises OutputFormatNotEnabled (discrete.zig:215).  What is written instead is
multi-block 2-D PLOT3D, whose grid file is exactly those planes: little-endian, no record markers,
    int32 nblocks | nblocks x (int32 ni, int32 nj) | per block: x[ni*nj] then y[ni*nj] (float64, i fastest)
and, for P,Q, a PLOT3D function file: int32 nblocks | nblocks x (int32 ni, nj, nvar=2) | per block: P plane, Q plane."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi


class OutputFormatNotEnabled(RuntimeError):
    """discrete.zig:215 error.OutputFormatNotEnabled"""


def block_planes(points) -> tuple[np.ndarray, np.ndarray]:
    """(x, y) planes of one block's (ni, nj, 2) array, element j*ni + i, transposed on the device (cgns.zig:75-104)."""
    xy = np.ascontiguousarray(points, dtype=np.float64)
    ni, nj = xy.shape[0], xy.shape[1]
    x = np.empty(ni * nj, dtype=np.float64)
    y = np.empty(ni * nj, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _capi.check(_capi.lib().tm_export_soa(xy.ctypes.data_as(dp), ni, nj, x.ctypes.data_as(dp), y.ctypes.data_as(dp)))
    return x, y


def write_plot3d(filename, sizes, planes):
    """sizes: [(ni, nj)], planes: [(x, y)] with i fastest.  Pure file writing (no device)."""
    with open(filename, "wb") as f:
        np.array([len(sizes)], dtype="<i4").tofile(f)
        np.array(sizes, dtype="<i4").reshape(-1, 2).tofile(f)
        for (ni, nj), (x, y) in zip(sizes, planes):
            assert x.size == ni * nj and y.size == ni * nj
            np.asarray(x, dtype="<f8").tofile(f)
            np.asarray(y, dtype="<f8").tofile(f)


def write_plot3d_function(filename, sizes, fields):
    """fields: per block a list of planes (here [P, Q])."""
    with open(filename, "wb") as f:
        np.array([len(sizes)], dtype="<i4").tofile(f)
        for (ni, nj), fl in zip(sizes, fields):
            np.array([ni, nj, len(fl)], dtype="<i4").tofile(f)
        for (ni, nj), fl in zip(sizes, fields):
            for p in fl:
                assert p.size == ni * nj
                np.asarray(p, dtype="<f8").tofile(f)


def read_plot3d(filename):
    """-> [(ni, nj, x(ni,nj), y(ni,nj))] with x[i, j] indexing restored."""
    with open(filename, "rb") as f:
        nb = int(np.fromfile(f, dtype="<i4", count=1)[0])
        sizes = np.fromfile(f, dtype="<i4", count=2 * nb).reshape(nb, 2)
        out = []
        for ni, nj in sizes:
            x = np.fromfile(f, dtype="<f8", count=ni * nj).reshape(nj, ni).T
            y = np.fromfile(f, dtype="<f8", count=ni * nj).reshape(nj, ni).T
            out.append((int(ni), int(nj), x, y))
    return out


def _format_of(filename):
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".cgns":
        raise OutputFormatNotEnabled("CGNS output needs the cgns library (reference: -Duse-cgns); use .xyz / .p3d (PLOT3D)")
    if ext not in (".xyz", ".p3d", ".x"):
        raise OutputFormatNotEnabled(f"unknown output format {ext!r}; supported: .xyz / .p3d / .x (PLOT3D)")
    return "plot3d"


def write_mesh(mesh, filename):
    """discrete.zig:197-216 Mesh.write"""
    _format_of(filename)
    sizes = [(b.points.data.shape[0], b.points.data.shape[1]) for b in mesh.blocks]
    write_plot3d(filename, sizes, [block_planes(b.points.data) for b in mesh.blocks])


def write_smoother(smoother, filename, with_control_function=True):
    """smooth.zig:396-414 RowCompressedMatrixSystem2d.write: coordinates resident in the handle + control function."""
    _format_of(filename)
    mesh = smoother._mesh
    L = _capi.lib()
    dp = C.POINTER(C.c_double)
    sizes, planes, fields = [], [], []
    for b, blk in enumerate(mesh.blocks):
        ni, nj = blk.points.data.shape[0], blk.points.data.shape[1]
        x, y, p, q = (np.empty(ni * nj, dtype=np.float64) for _ in range(4))
        _capi.check(L.tm_smoother_export_soa(smoother._h, b, x.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                             p.ctypes.data_as(dp) if with_control_function else None,
                                             q.ctypes.data_as(dp) if with_control_function else None))
        sizes.append((ni, nj))
        planes.append((x, y))
        fields.append([p, q])
    write_plot3d(filename, sizes, planes)
    if with_control_function:
        write_plot3d_function(os.path.splitext(filename)[0] + ".f", sizes, fields)
