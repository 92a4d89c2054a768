"""Mirror of reference src/core/templates/O4H.zig: the O4H blocking of a turbine cascade passage --
8 blocks (blade_up, blade_down, in, out, down, up, upstream, downstream), 21 connections (3 periodic with
(0, pitch)), inlet / outlet conditions.  Host-side (SURVEY.md "next" row N1); every block is seeded by
Block2d.init, i.e. TFI on the MI355X (or by the injected `tfi` callable in CPU-only tests)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .. import clustering as cluster
from ..boundary import Condition, ConditionTag, Connection, Range, Side
from ..discrete import Block2d, Edge, EdgeView, Mesh
from ..geometry import Line


@dataclass
class NumCells:
    """O4H.zig:46-66"""

    o_grid: int
    middle_i: int
    in_up_j: int
    in_down_j: int
    in_i: int
    out_up_j: int
    out_down_j: int
    out_i: int
    down_j: int
    bulge: int
    upstream_i: int
    downstream_i: int


class _SplineCurve:
    def __init__(self, spline):
        self.spline = spline

    def interpolate(self, u):
        return self.spline.interpolate(u)


def projectNormal(edge, distance):
    """O4H.zig:531-574: offset every point along the (right-hand) normal of the discrete tangent."""
    edge = np.asarray(edge, dtype=np.float64)
    n = len(edge)
    out = np.empty_like(edge)

    def proj(x_i, x_xi):
        a = np.sqrt(x_xi[0] * x_xi[0] + x_xi[1] * x_xi[1])   # types.abs
        s = 1.0 / a
        nrm = np.array([s * x_xi[1], s * (-x_xi[0])])
        return np.array([x_i[0] + distance * nrm[0], x_i[1] + distance * nrm[1]])

    for i in range(1, n - 1):
        out[i] = proj(edge[i], 0.5 * (edge[i + 1] - edge[i - 1]))
    out[0] = proj(edge[0], edge[1] - edge[0])
    out[n - 1] = proj(edge[n - 1], edge[n - 1] - edge[n - 2])
    return out


@dataclass
class O4H:
    """O4H.zig:38-66"""

    blade_clustering: object
    num_cells: NumCells
    inlet_distance: Optional[float] = None
    outlet_distance: Optional[float] = None

    def run(self, geom, tfi=None) -> Mesh:
        """O4H.zig:67-528"""
        nc = self.num_cells
        V = lambda a, b: np.array([a, b], dtype=np.float64)   # noqa: E731
        mk = (lambda a, b, c, d: Block2d.init(a, b, c, d)) if tfi is None else tfi

        def line_edge(n, start, end, cl=None):
            return Edge.init(n, Line(start, end), cl or cluster.Uniform())

        num_cells_up = nc.in_up_j + nc.middle_i + nc.bulge + nc.out_up_j + nc.out_i
        num_cells_down = nc.in_down_j + nc.middle_i + nc.out_down_j
        profile_length = geom.profile.up_part.total_length + geom.profile.down_part.total_length
        default_spacing = profile_length / float(num_cells_up + num_cells_down)

        down_edge = Edge.init(num_cells_down + 1, _SplineCurve(geom.profile.down_part), self.blade_clustering)
        up_edge = Edge.init(num_cells_up + 1, _SplineCurve(geom.profile.up_part), self.blade_clustering)
        leading_edge = up_edge.points[0].copy()
        down_edge.points[0] = leading_edge
        trailing_edge = up_edge.points[-1].copy()
        down_edge.points[-1] = trailing_edge

        inlet_distance = self.inlet_distance if self.inlet_distance is not None else default_spacing * float(nc.upstream_i)
        outlet_distance = self.outlet_distance if self.outlet_distance is not None else default_spacing * float(nc.downstream_i)

        d = 0.001   # O4H.zig:102
        down_outer_edge = Edge(projectNormal(down_edge.points, d), down_edge.clustering.copy())
        up_outer = projectNormal(up_edge.points, -d)
        up_outer[0] = down_outer_edge.points[0]
        up_outer[-1] = down_outer_edge.points[-1]
        up_outer_edge = Edge(up_outer, up_edge.clustering.copy())

        mesh = Mesh()
        tanh = cluster.SingleHyperbolicClustering(0.01)

        # Block BLADE_UP (0)
        blade_up_i_min, blade_up_i_max = up_edge, up_outer_edge
        blade_up_j_min = line_edge(nc.o_grid + 1, blade_up_i_min.points[0], blade_up_i_max.points[0], tanh)
        blade_up_j_max = line_edge(nc.o_grid + 1, blade_up_i_min.points[-1], blade_up_i_max.points[-1], tanh)
        mesh.addBlock("blade_up", mk(blade_up_i_min, blade_up_i_max, blade_up_j_min, blade_up_j_max))
        blade_up_id = 0

        # Block BLADE_DOWN (1)
        blade_down_i_min, blade_down_i_max = down_edge, down_outer_edge
        mesh.addBlock("blade_down", mk(blade_down_i_min, blade_down_i_max, blade_up_j_min, blade_up_j_max))
        blade_down_id = 1

        # Block IN (2)
        in_j_min = Edge.combine([EdgeView(blade_up_i_max, nc.in_up_j, 0), EdgeView(blade_down_i_max, 0, nc.in_down_j)])
        assert len(in_j_min.points) == nc.in_up_j + nc.in_down_j + 1
        in_x_00, in_x_01 = in_j_min.points[0], in_j_min.points[-1]
        in_x_start = leading_edge[0] - inlet_distance * 0.5
        in_x_10 = V(in_x_start, leading_edge[1] + geom.pitch * 0.25)
        in_x_11 = V(in_x_start, leading_edge[1] - geom.pitch * 0.25)
        in_j_max = line_edge(len(in_j_min.points), in_x_10, in_x_11)
        in_i_min = line_edge(nc.in_i + 1, in_x_00, in_x_10)
        in_i_max = line_edge(nc.in_i + 1, in_x_01, in_x_11)
        mesh.addBlock("in", mk(in_i_min, in_i_max, in_j_min, in_j_max))
        in_id = 2

        # Block OUT (3)
        n_down_pts, n_up_pts = len(blade_down_i_max.points), len(blade_up_i_max.points)
        out_j_min = Edge.combine([EdgeView(blade_down_i_max, nc.in_down_j + nc.middle_i, n_down_pts - 1),
                                  EdgeView(blade_up_i_max, n_up_pts - 1, nc.in_up_j + nc.bulge + nc.middle_i + nc.out_i)])
        assert len(out_j_min.points) == nc.out_down_j + nc.out_up_j + 1
        out_x_00, out_x_01 = out_j_min.points[0], out_j_min.points[-1]
        out_x_end = outlet_distance * 0.5 + trailing_edge[0]
        out_x_10 = V(out_x_end, trailing_edge[1] - geom.pitch * 0.25)
        out_x_11 = V(out_x_end, trailing_edge[1] + geom.pitch * 0.25)
        out_j_max = line_edge(len(out_j_min.points), out_x_10, out_x_11)
        out_i_min = line_edge(nc.out_i + 1, out_x_00, out_x_10)
        out_i_max = line_edge(nc.out_i + 1, out_x_01, out_x_11)
        mesh.addBlock("out", mk(out_i_min, out_i_max, out_j_min, out_j_max))
        out_id = 3

        # Block DOWN (4)
        down_i_min = Edge.combine([EdgeView(in_i_max, nc.in_i, 0), EdgeView(blade_down_i_max, nc.in_down_j, nc.in_down_j + nc.middle_i),
                                   EdgeView(out_i_min, 0, nc.out_i)])
        down_x_00 = in_x_11
        down_x_01 = leading_edge - V(0.0, 0.5 * geom.pitch)
        down_x_11 = trailing_edge - V(0.0, 0.5 * geom.pitch)
        down_x_10 = out_x_10
        down_i_max = line_edge(len(down_i_min.points), down_x_01, down_x_11)
        down_j_min = line_edge(nc.down_j + 1, down_x_00, down_x_01)
        down_j_max = line_edge(len(down_j_min.points), down_x_10, down_x_11)
        mesh.addBlock("down", mk(down_i_min, down_i_max, down_j_min, down_j_max))
        down_id = 4

        # Block UP (5)
        up_j_min = out_i_max
        up_i_min = Edge.combine([EdgeView(blade_up_i_max, nc.in_up_j + nc.middle_i + nc.bulge + nc.out_i, nc.in_up_j), EdgeView(in_i_min, 0, nc.in_i)])
        up_x_11 = leading_edge + V(0.0, 0.5 * geom.pitch)
        up_x_i_max_middle = trailing_edge + V(0.0, 0.5 * geom.pitch)
        up_x_01 = out_x_11
        up_x_10 = in_x_10
        up_i_max_0 = line_edge(nc.bulge + 1, up_x_01, up_x_i_max_middle)
        up_i_max_1 = line_edge(len(up_i_min.points) - nc.bulge, up_x_i_max_middle, up_x_11)
        up_i_max = Edge.combine([EdgeView(up_i_max_0, 0, nc.bulge), EdgeView(up_i_max_1, 0, len(up_i_max_1.points) - 1)])
        up_j_max = line_edge(nc.out_i + 1, up_x_10, up_x_11)
        mesh.addBlock("up", mk(up_i_min, up_i_max, up_j_min, up_j_max))
        up_id = 5

        # Block UPSTREAM (6)
        upstream_j_max = Edge.combine([EdgeView(down_j_min, nc.down_j, 0), EdgeView(in_j_max, len(in_j_max.points) - 1, 0),
                                       EdgeView(up_j_max, 0, len(up_j_max.points) - 1)])
        upstream_x_10, upstream_x_11 = upstream_j_max.points[0], upstream_j_max.points[-1]
        upstream_x_00 = V(leading_edge[0] - inlet_distance, leading_edge[1] - 0.5 * geom.pitch)
        upstream_x_01 = V(leading_edge[0] - inlet_distance, leading_edge[1] + 0.5 * geom.pitch)
        upstream_j_min = line_edge(len(upstream_j_max.points), upstream_x_00, upstream_x_01)
        upstream_i_min = line_edge(nc.upstream_i + 1, upstream_x_00, upstream_x_10)
        upstream_i_max = line_edge(nc.upstream_i + 1, upstream_x_01, upstream_x_11)
        mesh.addBlock("upstream", mk(upstream_i_min, upstream_i_max, upstream_j_min, upstream_j_max))
        upstream_id = 6

        # Block DOWNSTREAM (7)
        downstream_j_min = Edge.combine([EdgeView(down_j_max, len(down_j_max.points) - 1, 0), EdgeView(out_j_max, 0, len(out_j_max.points) - 1),
                                         EdgeView(up_i_max_0, 0, len(up_i_max_0.points) - 1)])
        downstream_x_00, downstream_x_01 = downstream_j_min.points[0], downstream_j_min.points[-1]
        downstream_x_10 = downstream_x_00 + V(outlet_distance, 0.0)
        downstream_x_11 = downstream_x_10 + V(0.0, geom.pitch)
        downstream_j_max = line_edge(len(downstream_j_min.points), downstream_x_10, downstream_x_11)
        downstream_i_min = line_edge(nc.downstream_i + 1, downstream_x_00, downstream_x_10)
        downstream_i_max = line_edge(nc.downstream_i + 1, downstream_x_01, downstream_x_11)
        mesh.addBlock("downstream", mk(downstream_i_min, downstream_i_max, downstream_j_min, downstream_j_max))
        downstream_id = 7

        # Connections (O4H.zig:423-515)
        n_in_j = len(in_j_min.points)
        n_up_i = len(up_i_min.points)
        n_down_i = len(down_i_min.points)
        n_ds_j = len(downstream_j_min.points)
        C = lambda r0, r1, per=None: mesh.connections.append(Connection((Range(*r0), Range(*r1)), per))   # noqa: E731
        S = Side
        C((blade_up_id, S.j_min, 0, nc.o_grid), (blade_down_id, S.j_min, 0, nc.o_grid))
        C((blade_up_id, S.j_max, 0, nc.o_grid), (blade_down_id, S.j_max, 0, nc.o_grid))
        C((down_id, S.j_min, nc.down_j, 0), (upstream_id, S.j_max, 0, nc.down_j))
        C((in_id, S.j_max, n_in_j - 1, 0), (upstream_id, S.j_max, nc.down_j, nc.down_j + n_in_j - 1))
        C((in_id, S.i_max, 0, nc.in_i), (down_id, S.i_min, nc.in_i, 0))
        C((up_id, S.j_max, 0, nc.out_i), (upstream_id, S.j_max, nc.down_j + n_in_j - 1, len(upstream_j_max.points) - 1))
        C((in_id, S.i_min, 0, nc.in_i), (up_id, S.i_min, n_up_i - nc.in_i - 1, n_up_i - 1))
        C((down_id, S.j_max, nc.down_j, 0), (downstream_id, S.j_min, 0, nc.down_j))
        C((out_id, S.j_max, 0, len(out_j_max.points) - 1), (downstream_id, S.j_min, nc.down_j, nc.down_j + len(out_j_max.points) - 1))
        C((out_id, S.i_min, 0, nc.out_i), (down_id, S.i_min, n_down_i - 1 - nc.out_i, n_down_i - 1))
        C((out_id, S.i_max, 0, nc.out_i), (up_id, S.j_min, 0, nc.out_i))
        C((up_id, S.i_max, 0, nc.bulge), (downstream_id, S.j_min, n_ds_j - 1 - nc.bulge, n_ds_j - 1))
        C((blade_up_id, S.i_max, 0, nc.in_up_j), (in_id, S.j_min, nc.in_up_j, 0))
        C((blade_up_id, S.i_max, nc.in_up_j, nc.in_up_j + nc.middle_i + nc.bulge + nc.out_i), (up_id, S.i_min, n_up_i - 1 - nc.in_i, 0))
        C((blade_up_id, S.i_max, nc.in_up_j + nc.bulge + nc.middle_i + nc.out_i, n_up_pts - 1), (out_id, S.j_min, len(out_j_min.points) - 1, nc.out_down_j))
        C((blade_down_id, S.i_max, 0, nc.in_down_j), (in_id, S.j_min, nc.in_up_j, n_in_j - 1))
        C((blade_down_id, S.i_max, nc.in_down_j, nc.in_down_j + nc.middle_i), (down_id, S.i_min, nc.in_i, n_down_i - 1 - nc.out_i))
        C((blade_down_id, S.i_max, nc.in_down_j + nc.middle_i, n_down_pts - 1), (out_id, S.j_min, 0, nc.out_down_j))
        per = (0.0, geom.pitch)
        C((upstream_id, S.i_min, 0, nc.upstream_i), (upstream_id, S.i_max, 0, nc.upstream_i), per)
        C((down_id, S.i_max, 0, len(down_i_max.points) - 1), (up_id, S.i_max, len(up_i_max.points) - 1, len(up_i_max.points) - len(down_i_max.points)), per)
        C((downstream_id, S.i_min, 0, nc.downstream_i), (downstream_id, S.i_max, 0, nc.downstream_i), per)

        # Boundary conditions (O4H.zig:518-521)
        mesh.boundary_conditions.append(Condition(Range(upstream_id, S.j_min, 0, len(upstream_j_min.points) - 1), ConditionTag.inlet))
        mesh.boundary_conditions.append(Condition(Range(downstream_id, S.j_max, 0, len(downstream_j_max.points) - 1), ConditionTag.outlet))
        return mesh
