"""K1 parity: TFI on the MI355X through the C-ABI vs the CPU oracle -- bit-exact (fp64, same
operation order, no FMA contraction).  Reference: src/core/tfi.zig:112-208, :19-67."""
import math

import numpy as np
import pytest

from oracle import oracle
from turbomesh_amd import TmError, clustering, configs, tfi
from turbomesh_amd.discrete import Block2d, Edge
from turbomesh_amd.types import Mat2d

pytestmark = pytest.mark.gpu


def _curved_edges(ni, nj, cl_i1, cl_i2, cl_j1, cl_j2):
    s1, s2, t1, t2 = cl_i1.compute(ni), cl_i2.compute(ni), cl_j1.compute(nj), cl_j2.compute(nj)
    for c in (s1, s2, t1, t2):   # the reference asserts first == 0 and last == 1.0 exactly (tfi.zig:135-145)
        c[0], c[-1] = 0.0, 1.0
    i_min = np.stack([s1 * 2.0, 0.15 * np.sin(math.pi * s1)], axis=1)
    i_max = np.stack([s2 * 2.0 + 0.1 * np.sin(math.pi * s2), 1.0 + 0.2 * np.sin(2 * math.pi * s2)], axis=1)
    j_min = np.stack([-0.1 * np.sin(math.pi * t1), t1], axis=1)
    j_max = np.stack([2.0 + 0.1 * np.sin(math.pi * t2), t2], axis=1)
    j_min[0], j_min[-1] = i_min[0], i_max[0]
    j_max[0], j_max[-1] = i_min[-1], i_max[-1]
    return Edge(i_min, s1), Edge(i_max, s2), Edge(j_min, t1), Edge(j_max, t2)


@pytest.mark.parametrize("ni,nj", [(3, 3), (2, 2), (17, 33), (64, 64), (65, 257), (300, 70), (5, 1000)])
def test_tfi_bit_exact_uniform(ni, nj):
    e = configs.single_block_edges(ni, nj)
    got = Block2d.init(*e).points.data
    ref = oracle.tfi_block(e[0].points, e[1].points, e[2].points, e[3].points, e[0].clustering, e[1].clustering, e[2].clustering, e[3].clustering)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("ni,nj", [(33, 41), (129, 130)])
def test_tfi_bit_exact_mixed_clusterings(ni, nj):
    # opposite edges clustered differently: exercises the boundary-blended u,v (tfi.zig:185-186)
    e = _curved_edges(ni, nj, clustering.Uniform(), clustering.Roberts(0.5, 1.05), clustering.SingleHyperbolicClustering(0.002), clustering.Roberts(0.0, 1.2))
    got = Block2d.init(*e).points.data
    ref = oracle.tfi_block(e[0].points, e[1].points, e[2].points, e[3].points, e[0].clustering, e[1].clustering, e[2].clustering, e[3].clustering)
    assert got.tobytes() == ref.tobytes()
    assert not np.isnan(got).any()


def test_tfi_commented_reference_kat():
    # tfi.zig:230-260 (commented test): rectangle [0,1]x[0,2], arclength clustering {0,.5,1}: centre = (0.5, 1.0)
    i_min = Edge(np.array([[0, 0], [0.5, 0], [1, 0]], float), np.array([0, 0.5, 1.0]))
    i_max = Edge(np.array([[0, 2], [0.5, 2], [1, 2]], float), np.array([0, 0.5, 1.0]))
    j_min = Edge(np.array([[0, 0], [0, 1], [0, 2]], float), np.array([0, 0.5, 1.0]))
    j_max = Edge(np.array([[1, 0], [1, 1], [1, 2]], float), np.array([0, 0.5, 1.0]))
    got = Block2d.init(i_min, i_max, j_min, j_max).points.data
    assert got[1, 1, 0] == 0.5 and got[1, 1, 1] == 1.0
    assert np.array_equal(got[:, 0], i_min.points) and np.array_equal(got[0, :], j_min.points)


def test_tfi_error_behaviour():
    e = configs.single_block_edges(9, 7)
    bad = e[0].clustering.copy()
    bad[-1] = 0.999   # tfi.zig:136 assert -> error code
    data = Mat2d((9, 7))
    with pytest.raises(TmError) as ei:
        tfi.linear2dBoundaryBlendedControlFunction(data, e[0].points, e[1].points, e[2].points, e[3].points, bad, e[1].clustering, e[2].clustering, e[3].clustering)
    assert ei.value.code == -6
    pts = e[2].points.copy()
    pts[0] += 1e-6   # corner mismatch beyond 1e-10 (tfi.zig:153)
    with pytest.raises(TmError) as ei:
        tfi.linear2dBoundaryBlendedControlFunction(data, e[0].points, e[1].points, pts, e[3].points, e[0].clustering, e[1].clustering, e[2].clustering, e[3].clustering)
    assert ei.value.code == -3
    with pytest.raises(TmError) as ei:   # inconsistent sizes
        tfi.linear2dBoundaryBlendedControlFunction(data, e[0].points[:-1], e[1].points, e[2].points, e[3].points, e[0].clustering, e[1].clustering, e[2].clustering, e[3].clustering)
    assert ei.value.code == -1
    assert np.isnan(data.data).all()   # untouched (Mat2d NaN init, types.zig:86)


def test_tfi_linear2d_bit_exact():
    e = _curved_edges(40, 77, clustering.Uniform(), clustering.Uniform(), clustering.Uniform(), clustering.Uniform())
    out = np.empty((40, 77, 2))
    tfi.linear2d(out, e[0].points, e[1].points, e[2].points, e[3].points)
    ref = oracle.tfi_linear2d(e[0].points, e[1].points, e[2].points, e[3].points)
    assert out.tobytes() == ref.tobytes()


def test_tfi_full_size_properties():
    # BASELINE config 2 size: 4096 x 4096 (256 MiB).  Size-independent properties instead of an oracle run:
    n = 4096
    e = configs.single_block_edges(n, n)
    got = Block2d.init(*e).points.data
    assert not np.isnan(got).any()
    # boundary = blend evaluated at the edge = edge point up to rounding
    assert np.abs(got[:, 0] - e[0].points).max() < 4e-16 and np.abs(got[:, -1] - e[1].points).max() < 4e-16
    assert np.abs(got[0, :] - e[2].points).max() < 4e-16 and np.abs(got[-1, :] - e[3].points).max() < 4e-16
    # equal opposite clusterings: u = s_i, v = t_j exactly -> x = s_i (both walls have x = s), closed form for y
    s, t = e[0].clustering, e[2].clustering
    assert np.abs(got[:, :, 0] - s[:, None]).max() < 1e-15
    y_expected = (1 - t)[None, :] * e[0].points[:, 1][:, None] + t[None, :] * e[1].points[:, 1][:, None]
    assert np.abs(got[:, :, 1] - y_expected).max() < 1e-15
    # sampled rows agree with the oracle bit for bit: a row of the blend depends only on its own edge
    # data, the j edges and the four corners, so the sub-problem with rows {0, ..., n-1} reproduces them
    rows = np.array([0, 1, 2, 777, 2048, 3333, 4094, 4095])
    sub = oracle.tfi_block(e[0].points[rows], e[1].points[rows], e[2].points, e[3].points, s[rows], s[rows], t, t)
    assert np.array_equal(got[rows], sub)
