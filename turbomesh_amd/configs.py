"""Synthetic workloads of the hot path built with the mirrored
reference API: deterministic edge functions -> Edge -> Block2d.init (TFI) -> Mesh with
connections / boundary conditions.

Every builder takes `tfi=`: the callable that seeds a block from its four edges.  The default
is the product path (turbomesh_amd.tfi on the MI355X); CPU-only tests inject the oracle's TFI.
After seeding, `snap` writes the exact edge points back into the block perimeter so the two
sides of a connection agree bit for bit (the reference's connectionDataCheck wants 1e-15,
smooth.zig:220-275)."""
from __future__ import annotations

import math

import numpy as np

from . import clustering as cluster
from .boundary import Condition, ConditionTag, Connection, Range, Side
from .discrete import Block2d, Edge, Mesh
from .types import Mat2d


def _default_tfi(i_min, i_max, j_min, j_max):
    return Block2d.init(i_min, i_max, j_min, j_max)


def _seed(i_min: Edge, i_max: Edge, j_min: Edge, j_max: Edge, tfi, snap=True) -> Block2d:
    blk = (tfi or _default_tfi)(i_min, i_max, j_min, j_max)
    if snap:
        d = blk.points.data
        d[:, 0] = i_min.points       # i_min edge: j = 0
        d[:, -1] = i_max.points      # i_max edge: j = nj-1
        d[0, :] = j_min.points       # j_min edge: i = 0
        d[-1, :] = j_max.points      # j_max edge: i = ni-1
    return blk


def _uniform_edge(points):
    return Edge(points, cluster.Uniform().compute(len(points)))


# ---------------------------------------------------------------------------------------------
# config 2: single block, sinusoidal walls (A = 0.1), uniform clustering, Laplace control function
# ---------------------------------------------------------------------------------------------
def single_block_edges(ni, nj=None, amplitude=0.1):
    nj = nj or ni
    s = cluster.Uniform().compute(ni)
    t = cluster.Uniform().compute(nj)
    a = amplitude
    i_min = np.stack([s, a * np.sin(2 * math.pi * s)], axis=1)            # j = 0
    i_max = np.stack([s, 1.0 - a * np.sin(2 * math.pi * s)], axis=1)      # j = nj-1
    j_min = np.stack([np.zeros(nj), t], axis=1)                           # i = 0
    j_max = np.stack([np.ones(nj), t], axis=1)                            # i = ni-1
    i_min[0], i_min[-1] = (0.0, 0.0), (1.0, 0.0)                          # corners set exactly
    i_max[0], i_max[-1] = (0.0, 1.0), (1.0, 1.0)
    return Edge(i_min, s), Edge(i_max, s.copy()), Edge(j_min, t), Edge(j_max, t.copy())


def single_block(ni, nj=None, amplitude=0.1, tfi=None, perturb=0.0, seed=12345) -> Mesh:
    """SURVEY 8d config 2 (ni = nj = 4096 for the headline).  perturb > 0 displaces interior nodes by
    perturb*h*(U-1/2) with a seeded stream (convergence tests)."""
    e = single_block_edges(ni, nj, amplitude)
    blk = _seed(*e, tfi)
    if perturb:
        rng = np.random.default_rng(seed)
        d = blk.points.data
        h = 1.0 / (d.shape[0] - 1)
        d[1:-1, 1:-1] += perturb * h * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
    m = Mesh()
    m.addBlock("block", blk)
    return m


# ---------------------------------------------------------------------------------------------
# config 4: strip of blocks stacked in the i direction (i <-> y, j <-> x), one interface per pair
# ---------------------------------------------------------------------------------------------
def slices(count, n, first=0, tfi=None, perturb=0.0) -> Mesh:
    """SURVEY 8d config 5 (BASELINE configs[4]): `count` independent n x n single-block problems -- the spanwise cuts of a
    pseudo-3D mesh.  Slice m has the config-2 edges with amplitude A = 0.05 + 0.001 m; no connections, so a handle smooths them
    concurrently (one batched launch per pass) and ranks need no communication ("replicas only")."""
    m = Mesh()
    for k in range(first, first + count):
        one = single_block(n, n, amplitude=0.05 + 0.001 * k, tfi=tfi, perturb=perturb, seed=12345 + k)
        m.addBlock(f"slice_{k}", one.blocks[0])
    return m


def strip(nblocks, ni, nj=None, amplitude=0.1, tfi=None, reverse_odd=False, only_blocks=None) -> Mesh:
    """Block k: j_min edge (i = 0) = C_k, j_max edge (i = ni-1) = C_{k+1},
    C_k(t) = (t, k + A (1 - 2k/nblocks) sin 2 pi t).  Connections
    {k, j_max, 0, nj-1} <-> {k+1, j_min, 0, nj-1}.  reverse_odd flips the j direction of odd blocks
    (their ranges then run nj-1 -> 0) to exercise reversed connections.  only_blocks: build coordinates only for
    these block ids (the blocks a rank owns); the others are size-only placeholders."""
    nj = nj or ni
    t = cluster.Uniform().compute(nj)
    s = cluster.Uniform().compute(ni)

    def curve(k):
        c = np.stack([t, k + amplitude * (1.0 - 2.0 * k / nblocks) * np.sin(2 * math.pi * t)], axis=1)
        c[0], c[-1] = (0.0, float(k)), (1.0, float(k))
        return c

    curves = {k: curve(k) for k in range(nblocks + 1) if only_blocks is None or k in only_blocks or (k - 1) in only_blocks}
    m = Mesh()
    for k in range(nblocks):
        if only_blocks is not None and k not in only_blocks:
            m.addBlock(f"block_{k}", Block2d(Mat2d.placeholder((ni, nj))))
            continue
        left = np.stack([np.zeros(ni), k + s], axis=1)       # x = 0
        right = np.stack([np.ones(ni), k + s], axis=1)       # x = 1
        left[0], left[-1] = curves[k][0], curves[k + 1][0]
        right[0], right[-1] = curves[k][-1], curves[k + 1][-1]
        flip = reverse_odd and (k % 2 == 1)
        if not flip:
            blk = _seed(_uniform_edge(left), _uniform_edge(right), _uniform_edge(curves[k]), _uniform_edge(curves[k + 1]), tfi)
        else:   # j runs from x = 1 to x = 0
            blk = _seed(_uniform_edge(right), _uniform_edge(left), _uniform_edge(curves[k][::-1].copy()),
                        _uniform_edge(curves[k + 1][::-1].copy()), tfi)
        m.addBlock(f"block_{k}", blk)
    for k in range(nblocks - 1):
        f0 = reverse_odd and (k % 2 == 1)
        f1 = reverse_odd and ((k + 1) % 2 == 1)
        r0 = Range(k, Side.j_max, nj - 1 if f0 else 0, 0 if f0 else nj - 1)
        r1 = Range(k + 1, Side.j_min, nj - 1 if f1 else 0, 0 if f1 else nj - 1)
        m.connections.append(Connection((r0, r1), None))
    return m


# ---------------------------------------------------------------------------------------------
# periodic channel: one block, i_min -> i_max self connection with periodicity (0, pitch),
# optional inlet / outlet sliding boundaries on j_min / j_max
# ---------------------------------------------------------------------------------------------
def periodic_channel(ni, nj, pitch=0.8, amplitude=0.08, sliding=True, tfi=None) -> Mesh:
    """One block, i <-> x, j <-> y.  The j = 0 and j = nj-1 edges are the same curve shifted by (0, pitch) and
    joined by a periodic self connection (i_min -> i_max, smooth.zig:522-559).  The outlet (i = ni-1) is a fixed
    wavy wall; the inlet (i = 0) is either fixed or an `inlet` condition (sliding rows: x kept, y = first interior y)."""
    s = cluster.Uniform().compute(ni)
    t = cluster.Uniform().compute(nj)
    wave = amplitude * np.sin(2 * math.pi * s) * s * (1 - s) * 4
    low = np.stack([s, wave], axis=1)                      # j = 0
    up = low.copy()                                        # j = nj-1 = low + (0, pitch) exactly
    up[:, 1] = low[:, 1] + pitch
    inlet = np.stack([np.zeros(nj), low[0, 1] + t * pitch], axis=1)
    outlet = np.stack([1.0 + 0.05 * np.sin(2 * math.pi * t), low[-1, 1] + t * pitch + 0.03 * np.sin(4 * math.pi * t)], axis=1)
    inlet[0], inlet[-1] = low[0], up[0]
    outlet[0], outlet[-1] = low[-1], up[-1]
    m = Mesh()
    m.addBlock("channel", _seed(_uniform_edge(low), _uniform_edge(up), _uniform_edge(inlet), _uniform_edge(outlet), tfi))
    m.connections.append(Connection((Range(0, Side.i_min, 0, ni - 1), Range(0, Side.i_max, 0, ni - 1)), (0.0, pitch)))
    if sliding:
        m.boundary_conditions.append(Condition(Range(0, Side.j_min, 0, nj - 1), ConditionTag.inlet))
    return m


# ---------------------------------------------------------------------------------------------
# 2 x 2 blocks meeting in one junction point; i <-> x, j <-> y inside every block
# ---------------------------------------------------------------------------------------------
def _warp(xi, eta, a):
    x = xi + a * np.sin(2 * math.pi * eta) * np.cos(0.5 * math.pi * xi)      # wavy left wall, also warps the interior lines
    y = eta + a * np.sin(2 * math.pi * xi) * (1.0 - 0.5 * eta)               # wavy bottom and top walls
    return np.stack([x, y], axis=1)


def two_by_two(n, m=None, warp=0.04, tfi=None) -> Mesh:
    """Blocks 0 (bottom-left), 1 (bottom-right), 2 (top-left), 3 (top-right) of the warped unit square.
    Connections 0-1 and 2-3 across x = 1/2 (j_max <-> j_min), 0-2 and 1-3 across y = 1/2 (i_max <-> i_min);
    the centre node is a junction of four corners (smooth.zig:1340-1514)."""
    m_ = m or n
    u = cluster.Uniform().compute(n)      # along i (x)
    w = cluster.Uniform().compute(m_)     # along j (y)
    xs = [0.0, 0.5, 1.0]
    ys = [0.0, 0.5, 1.0]
    hor = {}   # horizontal edges (parametrised by i): key (col, yline)
    ver = {}   # vertical edges (parametrised by j): key (xline, row)
    for c in range(2):
        for yl in range(3):
            xi = xs[c] + u * 0.5
            hor[(c, yl)] = _warp(xi, np.full(n, ys[yl]), warp)
    for xl in range(3):
        for r in range(2):
            eta = ys[r] + w * 0.5
            ver[(xl, r)] = _warp(np.full(m_, xs[xl]), eta, warp)
    # shared corner points computed once so all four incident edges agree bit for bit
    for xl in range(3):
        for yl in range(3):
            corner = _warp(np.array([xs[xl]]), np.array([ys[yl]]), warp)[0]
            for c in range(2):
                if xl == c:
                    hor[(c, yl)][0] = corner
                if xl == c + 1:
                    hor[(c, yl)][-1] = corner
            for r in range(2):
                if yl == r:
                    ver[(xl, r)][0] = corner
                if yl == r + 1:
                    ver[(xl, r)][-1] = corner
    mesh = Mesh()
    for r in range(2):
        for c in range(2):
            blk = _seed(_uniform_edge(hor[(c, r)]), _uniform_edge(hor[(c, r + 1)]), _uniform_edge(ver[(c, r)]), _uniform_edge(ver[(c + 1, r)]), tfi)
            mesh.addBlock(f"b{r}{c}", blk)
    ni, nj = n, m_
    mesh.connections.append(Connection((Range(0, Side.j_max, 0, nj - 1), Range(1, Side.j_min, 0, nj - 1)), None))
    mesh.connections.append(Connection((Range(2, Side.j_max, 0, nj - 1), Range(3, Side.j_min, 0, nj - 1)), None))
    mesh.connections.append(Connection((Range(0, Side.i_max, 0, ni - 1), Range(2, Side.i_min, 0, ni - 1)), None))
    mesh.connections.append(Connection((Range(1, Side.i_max, 0, ni - 1), Range(3, Side.i_min, 0, ni - 1)), None))
    return mesh


# ---------------------------------------------------------------------------------------------
# flat plate with leading edge: blocks 0 (upper) and 1 (lower) share their j_min edges (the line running
# upstream from the leading edge) -- the smallest mesh that satisfies the `white` control function's
# hard-coded layout (blocks 0,1 walls at j = 0, connection 0 = {0,j_min,0..} <-> {1,j_min,0..};
# wall_control_function.zig:72, 204-213)
# ---------------------------------------------------------------------------------------------
def plate(ni, nj, height=0.5, upstream=0.5, thickness=0.12, wall_cluster=None, tfi=None) -> Mesh:
    """Round-nosed plate of finite thickness; sign = +1 upper block, -1 lower block (mirror image)."""
    s = cluster.Uniform().compute(ni)
    t = (wall_cluster or cluster.Uniform()).compute(nj)
    mesh = Mesh()
    for sign in (1.0, -1.0):
        yw = sign * thickness * np.sqrt(s) * (1.0 - 0.5 * s)
        wall = np.stack([s, yw], axis=1)                                                   # i_min edge (j = 0): the wall
        ang = 0.5 * math.pi * (1.0 - s)
        outer = np.stack([1.0 - (1.0 + upstream) * np.sin(ang), sign * height * np.cos(ang) ** 0.5], axis=1)   # i_max edge
        outer[0], outer[-1] = (-upstream, 0.0), (1.0, sign * height)
        lead = np.stack([-upstream * t, np.zeros(nj)], axis=1)                            # j_min edge (i = 0): upstream line
        trail = np.stack([np.ones(nj), yw[-1] + (sign * height - yw[-1]) * t], axis=1)    # j_max edge (i = ni-1)
        wall[0] = (0.0, 0.0)
        lead[0], lead[-1] = wall[0], outer[0]
        trail[0], trail[-1] = wall[-1], outer[-1]
        mesh.addBlock("up" if sign > 0 else "down", _seed(Edge(wall, s), Edge(outer, s.copy()), Edge(lead, t), Edge(trail, t.copy()), tfi))
    mesh.connections.append(Connection((Range(0, Side.j_min, 0, nj - 1), Range(1, Side.j_min, 0, nj - 1)), None))
    return mesh


def block_from_array(arr) -> Block2d:
    return Block2d(Mat2d.from_array(arr))
