// ORACLE (test infrastructure) -- faithful CPU restatement of
// reference src/core/smoothing/smooth.zig (system assembly, fill, residual, copy-back).
// Every function cites the reference lines it follows.  Quirks are kept on purpose
// (SURVEY.md H7): swapped (Q,P) on non-periodic interface rows, residual = 4th power,
// row-kind assignment order.  Two documented deviations: (1) a mesh without connections
// is accepted and simply has no junction points (the reference underflows, smooth.zig:1364);
// (2) lhs_values is sized by the true nnz, not by the reference's over-allocated capacity
// (smooth.zig:320-332) -- the arithmetic is unaffected.
#include "orc_system.hpp"
#include <algorithm>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------- smooth.zig:171-216
StencilData StencilData::init(Vec2d a_im1_j, Vec2d a_ip1_j, Vec2d a_i_jm1, Vec2d a_i_jp1, Float P, Float Q) {
    const Float x_xi = 0.5 * (a_ip1_j.data[0] - a_im1_j.data[0]);
    const Float x_eta = 0.5 * (a_i_jp1.data[0] - a_i_jm1.data[0]);
    const Float y_xi = 0.5 * (a_ip1_j.data[1] - a_im1_j.data[1]);
    const Float y_eta = 0.5 * (a_i_jp1.data[1] - a_i_jm1.data[1]);

    const Float g22 = x_eta * x_eta + y_eta * y_eta;
    const Float g12 = x_xi * x_eta + y_xi * y_eta;
    const Float g11 = x_xi * x_xi + y_xi * y_xi;

    StencilData s;
    s.data[i_j] = -2.0 * g22 - 2.0 * g11;
    s.data[ip1_j] = g22 * (1 + 0.5 * P);
    s.data[im1_j] = g22 * (1 - 0.5 * P);
    s.data[i_jp1] = g11 * (1 + 0.5 * Q);
    s.data[i_jm1] = g11 * (1 - 0.5 * Q);
    s.data[ip1_jp1] = -0.5 * g12;
    s.data[ip1_jm1] = 0.5 * g12;
    s.data[im1_jp1] = 0.5 * g12;
    s.data[im1_jm1] = -0.5 * g12;
    return s;
}

// ---------------------------------------------------------------- smooth.zig:220-275
void connectionDataCheck(const Mesh& mesh) {
    const Float abs_tol = 1e-15;
    for (Index c = 0; c < mesh.connections.size(); ++c) {
        const Connection& conn = mesh.connections[c];
        if (conn.ranges[0].len() != conn.ranges[1].len())
            throw Error(ORC_E_MISMATCH, "connection " + std::to_string(c) + ": ranges differ in length");
        RangeIterator it0 = conn.ranges[0].iterate(mesh), it1 = conn.ranges[1].iterate(mesh);
        Index p0, p1, k = 0;
        while (it0.next(p0)) {
            if (!it1.next(p1)) throw Error(ORC_E_MISMATCH, "non matching connection data");
            Vec2d x0 = mesh.blocks[conn.ranges[0].block].pts[p0];
            if (conn.has_periodicity) x0 = add(x0, conn.periodicity);
            const Vec2d x1 = mesh.blocks[conn.ranges[1].block].pts[p1];
            if (!eqlApprox(x0, x1, abs_tol))
                throw Error(ORC_E_MISMATCH,
                            "non matching points for connection " + std::to_string(c) + " point " + std::to_string(k));
            k += 1;
        }
    }
}

// ---------------------------------------------------------------- smooth.zig:1516-1522
static void appendIfUnique(std::vector<OverlappingPoint>& pts, Index id, Vec2d periodicity) {
    for (const auto& p : pts)
        if (p.global_id == id) return;
    if (pts.size() >= 4) throw Error(ORC_E_OVERFLOW, "junction point with more than 4 overlapping points");
    pts.push_back({id, periodicity});
}

// ---------------------------------------------------------------- smooth.zig:1340-1514
static std::vector<LaplacianPoint> initLaplacianPoints(const IndexConverter& ic, const Mesh& mesh) {
    std::vector<LaplacianPoint> lps;
    const Index nconn = mesh.connections.size();
    if (nconn == 0) return lps;   // deviation (1): reference underflows at :1364

    std::vector<Index> endpoint_ids(nconn * 4);   // :1346-1356
    for (Index c = 0; c < nconn; ++c) {
        const Connection& conn = mesh.connections[c];
        Index l0[2] = {0, 0}, l1[2] = {0, 0};
        conn.ranges[0].endpoints(mesh, l0);
        conn.ranges[1].endpoints(mesh, l1);
        endpoint_ids[c * 4 + 0] = ic.globalIndex(conn.ranges[0].block, l0[0]);
        endpoint_ids[c * 4 + 1] = ic.globalIndex(conn.ranges[1].block, l1[0]);
        endpoint_ids[c * 4 + 2] = ic.globalIndex(conn.ranges[0].block, l0[1]);
        endpoint_ids[c * 4 + 3] = ic.globalIndex(conn.ranges[1].block, l1[1]);
    }
    auto conn_periodicity = [&](Index connection_id) {
        const Connection& conn = mesh.connections[connection_id];
        return conn.has_periodicity ? conn.periodicity : vinit(0, 0);
    };

    const Index n = endpoint_ids.size();
    for (Index e = 0; e + 1 < n; ++e) {   // :1364
        const Index endpoint = endpoint_ids[e];
        for (Index chk = e + 1; chk < n; ++chk) {
            if (endpoint_ids[chk] != endpoint) continue;
            bool existing_point_found = false;   // :1371-1387
            for (auto& lp : lps) {
                const Index n0 = lp.overlapping_points.size();   // slice captured before any append
                for (Index k = 0; k < n0; ++k) {
                    if (lp.overlapping_points[k].global_id == endpoint) {
                        existing_point_found = true;
                        const Index to_add = (chk % 2 == 0) ? chk + 1 : chk - 1;
                        appendIfUnique(lp.overlapping_points, endpoint_ids[to_add], conn_periodicity(to_add / 4));
                    }
                }
            }
            if (!existing_point_found) {   // :1389-1436
                const Index p0 = e / 2, p1 = chk / 2;
                if (p0 == p1) throw Error(ORC_E_TOPOLOGY, "degenerate connection endpoint pair");
                LaplacianPoint lp;
                lp.rhs = vinit(0, 0);
                lp.overlapping_points.push_back({endpoint_ids[p0 * 2], vinit(0, 0)});
                lp.overlapping_points.push_back({endpoint_ids[p0 * 2 + 1], conn_periodicity(p0 / 2)});
                const Vec2d per1 = conn_periodicity(p1 / 2);
                appendIfUnique(lp.overlapping_points, endpoint_ids[p1 * 2], per1);
                appendIfUnique(lp.overlapping_points, endpoint_ids[p1 * 2 + 1], per1);
                lps.push_back(lp);
            }
        }
    }

    for (auto& lp : lps)   // :1442-1448
        std::stable_sort(lp.overlapping_points.begin(), lp.overlapping_points.end(),
                         [](const OverlappingPoint& a, const OverlappingPoint& b) { return a.global_id < b.global_id; });
    std::stable_sort(lps.begin(), lps.end(), [](const LaplacianPoint& a, const LaplacianPoint& b) {   // :1451-1455
        return a.overlapping_points[0].global_id < b.overlapping_points[0].global_id;
    });

    for (auto& lp : lps) {   // :1458-1511
        lp.stencil_ids.push_back(static_cast<int32_t>(lp.globalId()));
        for (const auto& op : lp.overlapping_points) {
            Index block, local, pi, pj;
            ic.localIndex(op.global_id, block, local);
            ic.index2d(block, local, pi, pj);
            const Block& b = mesh.blocks[block];
            Index pts[2][2];
            int npts = 0;
            auto push = [&](Index i, Index j) { pts[npts][0] = i; pts[npts][1] = j; npts += 1; };
            if (pi == 0) {
                if (pj == 0) push(1, 1);
                else if (pj == b.nj - 1) push(1, b.nj - 2);
                else { push(1, pj - 1); push(1, pj + 1); }
            } else if (pi == b.ni - 1) {
                if (pj == 0) push(b.ni - 2, 1);
                else if (pj == b.nj - 1) push(b.ni - 2, b.nj - 2);
                else { push(b.ni - 2, pj - 1); push(b.ni - 2, pj + 1); }
            } else {
                if (pj == 0) { push(pi - 1, 1); push(pi + 1, 1); }
                else if (pj == b.nj - 1) { push(pi - 1, pj - 1); push(pi + 1, pj - 1); }
                else throw Error(ORC_E_TOPOLOGY, "junction point is not a boundary point");
            }
            for (int k = 0; k < npts; ++k) {
                if (lp.stencil_ids.size() >= 6) throw Error(ORC_E_OVERFLOW, "junction stencil with more than 6 ids");
                lp.stencil_ids.push_back(static_cast<int32_t>(ic.globalIndex(block, b.index(pts[k][0], pts[k][1]))));
                lp.rhs = add(lp.rhs, op.periodicity);   // Vec2d.add, :1504
            }
        }
        std::sort(lp.stencil_ids.begin(), lp.stencil_ids.end());
    }
    return lps;
}

// ---------------------------------------------------------------- smooth.zig:1234-1332
void BlockBoundaryPoints::init(const IndexConverter& ic, const Mesh& mesh) {
    index_converter.init(mesh);
    laplacian_points = initLaplacianPoints(ic, mesh);
    kind.assign(index_converter.total, fixed);   // :1243

    auto bufferOf = [&](Index global_id) {
        Index block, local, i, j;
        ic.localIndex(global_id, block, local);
        ic.index2d(block, local, i, j);
        return index_converter.bufferIndex(block, i, j);
    };
    auto bufferOfLocal = [&](Index block, Index local) {
        Index i, j;
        ic.index2d(block, local, i, j);
        return index_converter.bufferIndex(block, i, j);
    };

    for (const auto& lp : laplacian_points) {   // :1246-1263
        kind[bufferOf(lp.overlapping_points[0].global_id)] = laplacian_smoothed;
        for (Index k = 1; k < lp.overlapping_points.size(); ++k) kind[bufferOf(lp.overlapping_points[k].global_id)] = connected;
    }

    for (const auto& bc : mesh.boundary_conditions) {   // :1265-1277
        if (bc.kind == inlet || bc.kind == outlet) {
            RangeIterator it = bc.range.iterate(mesh);
            Index local;
            while (it.next(local)) kind[bufferOfLocal(bc.range.block, local)] = sliding_circ;
        }
    }

    for (const auto& conn : mesh.connections) {   // :1280-1329
        RangeIterator it0 = conn.ranges[0].iterate(mesh), it1 = conn.ranges[1].iterate(mesh);
        Index l0, l1;
        auto endpoint = [&]() {
            it0.next(l0);
            it1.next(l1);
            const Index b0 = bufferOfLocal(conn.ranges[0].block, l0), b1 = bufferOfLocal(conn.ranges[1].block, l1);
            if (kind[b0] == fixed || kind[b0] == sliding_circ) kind[b1] = connected;
        };
        endpoint();
        // `for (0..connected_points.data[0].count)` (:1303): count is len-2 after the first
        // next(), so exactly the interior points; the final next() is the second end point.
        const Index middle = it0.count;
        for (Index k = 0; k < middle; ++k) {
            it0.next(l0);
            it1.next(l1);
            kind[bufferOfLocal(conn.ranges[0].block, l0)] = smoothed;
            kind[bufferOfLocal(conn.ranges[1].block, l1)] = connected;
        }
        endpoint();
    }
}

// ---------------------------------------------------------------- smooth.zig:518-616
void System::computeConnectionStencilPositions(const Connection& c, const RangeFillMatrixIterator& it, Index pos[9]) {
    if (c.ranges[0].block == c.ranges[1].block) {
        if (c.ranges[0].side == i_min && c.ranges[1].side == i_max) {
            if (it.in_connection_direction_shift[0] > 0) {
                const Index p[9] = {1, 4, 7, 0, 3, 6, 2, 5, 8};
                std::memcpy(pos, p, sizeof(p));
            } else {
                const Index p[9] = {7, 4, 1, 6, 3, 0, 8, 5, 2};
                std::memcpy(pos, p, sizeof(p));
            }
            return;
        }
        throw Error(ORC_E_TOPOLOGY, "same-block connection must be i_min -> i_max (smooth.zig:522-559)");
    }
    if (!(c.ranges[0].block < c.ranges[1].block)) throw Error(ORC_E_TOPOLOGY, "connection ranges[0].block must be < ranges[1].block");
    const int d0 = it.in_connection_direction_shift[0] > 0 ? 1 : -1;
    const int d1 = it.in_connection_direction_shift[1] > 0 ? 1 : -1;
    auto I = [](int v) { return static_cast<Index>(v); };
    switch (c.ranges[0].side) {
        case i_min:
            pos[0] = I(3 - 2 * d0); pos[1] = 3; pos[2] = I(3 + 2 * d0);
            pos[3] = I(2 - 2 * d0); pos[4] = 2; pos[5] = I(2 + 2 * d0);
            break;
        case i_max:
            pos[0] = I(2 - 2 * d0); pos[1] = 2; pos[2] = I(2 + 2 * d0);
            pos[3] = I(3 - 2 * d0); pos[4] = 3; pos[5] = I(3 + 2 * d0);
            break;
        case j_min:
            pos[0] = I(4 - d0); pos[1] = 4; pos[2] = I(4 + d0);
            pos[3] = I(1 - d0); pos[4] = 1; pos[5] = I(1 + d0);
            break;
        case j_max:
            pos[0] = I(1 - d0); pos[1] = 1; pos[2] = I(1 + d0);
            pos[3] = I(4 - d0); pos[4] = 4; pos[5] = I(4 + d0);
            break;
    }
    pos[6] = I(7 - d1); pos[7] = 7; pos[8] = I(7 + d1);
}

// ---------------------------------------------------------------- smooth.zig:421-778
void System::initNonZeroMatrixEntries() {
    std::vector<int32_t>& nz = lhs_i;
    nz.clear();
    lhs_p.assign(1, 0);
    const BlockBoundaryPoints& bp = boundary_points;

    // --- point based pass (:460-516) ---
    Index boundary_point_idx = 0;
    int32_t row_idx = 0;
    Index laplacian_count = 0;
    auto boundaryPoint = [&]() {   // :421-458
        switch (bp.kind[boundary_point_idx]) {
            case fixed: nz.push_back(row_idx); break;
            case smoothed: nz.insert(nz.end(), 9, -1); break;
            case connected: nz.insert(nz.end(), 2, -1); break;
            case laplacian_smoothed: {
                const auto& ids = bp.laplacian_points[laplacian_count].stencil_ids;
                nz.insert(nz.end(), ids.begin(), ids.end());
                laplacian_count += 1;
                break;
            }
            case sliding_circ: nz.insert(nz.end(), 2, -1); break;
        }
        boundary_point_idx += 1;
        lhs_p.push_back(static_cast<int32_t>(nz.size()));
        row_idx += 1;
    };
    for (const Block& block : mesh.blocks) {
        const int32_t col_size = static_cast<int32_t>(block.nj);
        for (Index j = 0; j < block.nj; ++j) boundaryPoint();   // edge j_min (i == 0)
        for (Index i = 1; i + 1 < block.ni; ++i) {
            boundaryPoint();   // edge i_min (j == 0)
            for (Index j = 1; j + 1 < block.nj; ++j) {
                nz.push_back(row_idx - col_size - 1);
                nz.push_back(row_idx - col_size);
                nz.push_back(row_idx - col_size + 1);
                nz.push_back(row_idx - 1);
                nz.push_back(row_idx);
                nz.push_back(row_idx + 1);
                nz.push_back(row_idx + col_size - 1);
                nz.push_back(row_idx + col_size);
                nz.push_back(row_idx + col_size + 1);
                lhs_p.push_back(static_cast<int32_t>(nz.size()));
                row_idx += 1;
            }
            boundaryPoint();   // edge i_max (j == nj-1)
        }
        for (Index j = 0; j < block.nj; ++j) boundaryPoint();   // edge j_max (i == ni-1)
    }

    // --- junction points: all but the lowest id are connected to it (:738-747) ---
    for (const auto& lp : bp.laplacian_points) {
        const Index smoothed_id = lp.globalId();
        for (Index k = 1; k < lp.overlapping_points.size(); ++k) {
            const Index s = nzStart(lp.overlapping_points[k].global_id);
            nz[s] = static_cast<int32_t>(smoothed_id);
            nz[s + 1] = static_cast<int32_t>(lp.overlapping_points[k].global_id);
        }
    }

    // --- connection based pass (:618-721) ---
    auto connectionEndpoint = [&](const Index local_ids[2], const Connection& conn) {   // :695-721
        Index i, j;
        index_converter.index2d(conn.ranges[0].block, local_ids[0], i, j);
        const Index buffer_id = bp.index_converter.bufferIndex(conn.ranges[0].block, i, j);
        switch (bp.kind[buffer_id]) {
            case fixed:
            case sliding_circ: {
                const int32_t g0 = static_cast<int32_t>(index_converter.globalIndex(conn.ranges[0].block, local_ids[0]));
                const int32_t g1 = static_cast<int32_t>(index_converter.globalIndex(conn.ranges[1].block, local_ids[1]));
                if (!(g0 < g1)) throw Error(ORC_E_TOPOLOGY, "connection end point ids not ascending");
                const Index s = nzStart(static_cast<Index>(g1));
                nz[s] = g0;
                nz[s + 1] = g1;
                break;
            }
            case laplacian_smoothed:
            case connected: break;
            default: throw Error(ORC_E_TOPOLOGY, "connection end point is a smoothed point (smooth.zig:719)");
        }
    };
    for (const Connection& conn : mesh.connections) {
        if (!(conn.ranges[0].block <= conn.ranges[1].block)) throw Error(ORC_E_TOPOLOGY, "ranges[0].block > ranges[1].block");
        if (!(conn.len() > 2 && conn.lenInternal() > 3)) throw Error(ORC_E_TOPOLOGY, "connection needs more than 3 internal points");
        RangeFillMatrixIterator it = RangeFillMatrixIterator::init(conn, mesh);
        Index pos[9];
        computeConnectionStencilPositions(conn, it, pos);
        Index ids[2];
        it.next(ids);
        connectionEndpoint(ids, conn);
        const Index middle = it.count - 1;
        for (Index k = 0; k < middle; ++k) {
            it.next(ids);
            const int32_t g0 = static_cast<int32_t>(index_converter.globalIndex(conn.ranges[0].block, ids[0]));
            const int32_t g1 = static_cast<int32_t>(index_converter.globalIndex(conn.ranges[1].block, ids[1]));
            {   // connect 2nd point to 1st
                if (!(g0 < g1)) throw Error(ORC_E_TOPOLOGY, "connected point ids not ascending");
                const Index s = nzStart(static_cast<Index>(g1));
                nz[s] = g0;
                nz[s + 1] = g1;
            }
            {   // smooth 1st point
                const Index s = nzStart(static_cast<Index>(g0));
                const int32_t dir0 = it.in_connection_direction_shift[0], dir1 = it.in_connection_direction_shift[1];
                const int32_t fi0 = it.first_internal_point_shift[0], fi1 = it.first_internal_point_shift[1];
                nz[s + pos[0]] = g0 - dir0 + fi0;
                nz[s + pos[1]] = g0 + fi0;
                nz[s + pos[2]] = g0 + dir0 + fi0;
                nz[s + pos[3]] = g0 - dir0;
                nz[s + pos[4]] = g0;
                nz[s + pos[5]] = g0 + dir0;
                nz[s + pos[6]] = g1 - dir1 + fi1;
                nz[s + pos[7]] = g1 + fi1;
                nz[s + pos[8]] = g1 + dir1 + fi1;
                for (int q = 0; q < 8; ++q)   // :679-687
                    if (nz[s + q] >= nz[s + q + 1]) throw Error(ORC_E_TOPOLOGY, "connection stencil columns not ascending");
            }
        }
        connectionEndpoint(it.position, conn);
    }

    // --- sliding boundary rows (:751-777) ---
    for (const Condition& bc : mesh.boundary_conditions) {
        if (!(bc.kind == inlet || bc.kind == outlet)) throw Error(ORC_E_TOPOLOGY, "wall condition in boundary_conditions (smooth.zig:775)");
        const int32_t shift = bc.range.firstInternalPointShift(mesh);
        RangeIterator it = bc.range.iterate(mesh);
        Index local;
        while (it.next(local)) {
            Index i, j;
            index_converter.index2d(bc.range.block, local, i, j);
            const Index boundary_id = bp.index_converter.bufferIndex(bc.range.block, i, j);
            if (bp.kind[boundary_id] != sliding_circ) continue;
            const int32_t g = static_cast<int32_t>(index_converter.globalIndex(bc.range.block, local));
            const Index s = nzStart(static_cast<Index>(g));
            if (shift > 0) {
                nz[s] = g;
                nz[s + 1] = g + shift;
            } else {
                nz[s] = g + shift;
                nz[s + 1] = g;
            }
        }
    }
}

// ---------------------------------------------------------------- smooth.zig:780-921
void System::initBoundaryData() {
    const BlockBoundaryPoints& bp = boundary_points;
    Index row_idx = 0, boundary_point_idx = 0, laplacian_count = 0;
    auto boundaryPoint = [&](const Block& block, Index point_idx) {   // :780-865
        const Index e = nzStart(row_idx);
        switch (bp.kind[boundary_point_idx]) {
            case fixed:
                lhs_values[e] = 1;
                rhs_x[row_idx] = block.pts[point_idx].data[0];
                rhs_y[row_idx] = block.pts[point_idx].data[1];
                break;
            case smoothed:
                rhs_x[row_idx] = 0;
                rhs_y[row_idx] = 0;
                break;
            case connected:
                lhs_values[e] = 1;
                lhs_values[e + 1] = -1;
                rhs_x[row_idx] = 0;
                rhs_y[row_idx] = 0;
                break;
            case laplacian_smoothed: {
                const LaplacianPoint& lp = bp.laplacian_points[laplacian_count];
                const int32_t id = static_cast<int32_t>(lp.globalId());
                Index position = 0;
                for (int32_t g : lp.stencil_ids) {
                    if (g == id) break;
                    position += 1;
                }
                for (Index k = 0; k < lp.stencil_ids.size(); ++k) lhs_values[e + k] = 1;
                lhs_values[e + position] = -static_cast<Float>(lp.stencil_ids.size()) + 1;
                rhs_x[row_idx] = 0;
                rhs_y[row_idx] = 0;
                laplacian_count += 1;
                break;
            }
            case sliding_circ: {
                // coefficients are rewritten before every x / y solve (fillXSpecific / fillYSpecific)
                Index b, l;
                index_converter.localIndex(row_idx, b, l);
                rhs_x[row_idx] = mesh.blocks[b].pts[l].data[0];
                rhs_y[row_idx] = 0.0;
                break;
            }
        }
        row_idx += 1;
        boundary_point_idx += 1;
    };
    for (const Block& block : mesh.blocks) {   // :873-901
        Index point_idx = 0;
        for (Index j = 0; j < block.nj; ++j) boundaryPoint(block, point_idx++);
        for (Index i = 1; i + 1 < block.ni; ++i) {
            boundaryPoint(block, point_idx++);
            for (Index j = 1; j + 1 < block.nj; ++j) {
                point_idx += 1;
                row_idx += 1;
            }
            boundaryPoint(block, point_idx++);
        }
        for (Index j = 0; j < block.nj; ++j) boundaryPoint(block, point_idx++);
    }
    for (const Connection& conn : mesh.connections) {   // :904-915
        if (!conn.has_periodicity) continue;
        RangeFillMatrixIterator it = RangeFillMatrixIterator::init(conn, mesh);
        Index ids[2];
        while (it.next(ids)) {
            const Index g1 = row_start[conn.ranges[1].block] + ids[1];
            rhs_x[g1] = -conn.periodicity.data[0];
            rhs_y[g1] = -conn.periodicity.data[1];
        }
    }
    for (const auto& lp : bp.laplacian_points) {   // :917-920
        rhs_x[lp.globalId()] = lp.rhs.data[0];
        rhs_y[lp.globalId()] = lp.rhs.data[1];
    }
}

// ---------------------------------------------------------------- smooth.zig:309-385
void System::init(const Mesh& m, int cf_algo, White w) {
    mesh = m;
    connectionDataCheck(mesh);
    for (const Block& b : mesh.blocks)
        if (b.ni < 3 || b.nj < 3) throw Error(ORC_E_SIZE, "block smaller than 3x3");
    dof = 0;
    row_start.resize(mesh.blocks.size());
    for (Index b = 0; b < mesh.blocks.size(); ++b) {
        row_start[b] = dof;
        dof += mesh.blocks[b].dof();
    }
    if (dof > static_cast<Index>(INT32_MAX) / 9) throw Error(ORC_E_SIZE, "dof exceeds c_int CSR index range");
    x_new.assign(dof, 0.0);
    y_new.assign(dof, 0.0);
    rhs_x.assign(dof, 0.0);
    rhs_y.assign(dof, 0.0);
    index_converter.init(mesh);
    boundary_points.init(index_converter, mesh);
    control_function.init(dof, mesh, cf_algo, w);
    initNonZeroMatrixEntries();
    lhs_values.assign(lhs_i.size(), 0.0);
    initBoundaryData();
    seeded_initial_guess = false;
}

// ---------------------------------------------------------------- smooth.zig:923-992
void System::fillBlockInternalPointData() {
    Index row_idx = 0;
    for (const Block& block : mesh.blocks) {
        Index point_idx = 0;
        point_idx += block.nj;
        row_idx += block.nj;
        for (Index i = 1; i + 1 < block.ni; ++i) {
            row_idx += 1;
            point_idx += 1;
            for (Index j = 1; j + 1 < block.nj; ++j) {
                const Vec2d im1_j = block.pts[point_idx - block.nj];
                const Vec2d i_jm1 = block.pts[point_idx - 1];
                const Vec2d i_jp1 = block.pts[point_idx + 1];
                const Vec2d ip1_j = block.pts[point_idx + block.nj];
                const Vec2d cf = control_function.data[row_idx];
                const StencilData st = StencilData::init(im1_j, ip1_j, i_jm1, i_jp1, cf.data[0], cf.data[1]);
                const Index e = nzStart(row_idx);
                lhs_values[e + 0] = st.get(StencilData::im1_jm1);
                lhs_values[e + 1] = st.get(StencilData::im1_j);
                lhs_values[e + 2] = st.get(StencilData::im1_jp1);
                lhs_values[e + 3] = st.get(StencilData::i_jm1);
                lhs_values[e + 4] = st.get(StencilData::i_j);
                lhs_values[e + 5] = st.get(StencilData::i_jp1);
                lhs_values[e + 6] = st.get(StencilData::ip1_jm1);
                lhs_values[e + 7] = st.get(StencilData::ip1_j);
                lhs_values[e + 8] = st.get(StencilData::ip1_jp1);
                rhs_x[row_idx] = 0;
                rhs_y[row_idx] = 0;
                point_idx += 1;
                row_idx += 1;
            }
            row_idx += 1;
            point_idx += 1;
        }
        row_idx += block.nj;
    }
}

// ---------------------------------------------------------------- smooth.zig:994-1105
void System::fillBlockConnectionData() {
    for (const Connection& conn : mesh.connections) {
        const Vec2d* pd0 = mesh.blocks[conn.ranges[0].block].pts;
        const Vec2d* pd1 = mesh.blocks[conn.ranges[1].block].pts;
        RangeFillMatrixIterator it = RangeFillMatrixIterator::init(conn, mesh);
        it.limitToRangeInternalPoints();
        Index pos[9];
        computeConnectionStencilPositions(conn, it, pos);
        Index ids[2];
        while (it.next(ids)) {
            const std::ptrdiff_t p0 = static_cast<std::ptrdiff_t>(ids[0]), p1 = static_cast<std::ptrdiff_t>(ids[1]);
            const Index g0 = row_start[conn.ranges[0].block] + ids[0];
            const Index e = nzStart(g0);
            const Vec2d im1_j = pd0[p0 - it.in_connection_direction_shift[0]];
            const Vec2d i_jm1 = pd0[p0 + it.first_internal_point_shift[0]];
            const Vec2d ip1_j = pd0[p0 + it.in_connection_direction_shift[0]];
            Vec2d i_jp1 = pd1[p1 + it.first_internal_point_shift[1]];
            const Vec2d cf = control_function.data[g0];
            StencilData st;
            if (conn.has_periodicity) {
                i_jp1 = add(i_jp1, negate(conn.periodicity));                                      // :1032
                st = StencilData::init(im1_j, ip1_j, i_jm1, i_jp1, cf.data[0], cf.data[1]);       // :1035-1042
            } else {
                st = StencilData::init(im1_j, ip1_j, i_jm1, i_jp1, cf.data[1], cf.data[0]);       // :1077-1084 (Q,P swapped)
            }
            lhs_values[e + pos[0]] = st.get(StencilData::im1_jm1);
            lhs_values[e + pos[1]] = st.get(StencilData::i_jm1);
            lhs_values[e + pos[2]] = st.get(StencilData::ip1_jm1);
            lhs_values[e + pos[3]] = st.get(StencilData::im1_j);
            lhs_values[e + pos[4]] = st.get(StencilData::i_j);
            lhs_values[e + pos[5]] = st.get(StencilData::ip1_j);
            lhs_values[e + pos[6]] = st.get(StencilData::im1_jp1);
            lhs_values[e + pos[7]] = st.get(StencilData::i_jp1);
            lhs_values[e + pos[8]] = st.get(StencilData::ip1_jp1);
            if (conn.has_periodicity) {   // :1060-1061
                const Float c = st.get(StencilData::im1_jp1) + st.get(StencilData::i_jp1) + st.get(StencilData::ip1_jp1);
                rhs_x[g0] = conn.periodicity.data[0] * c;
                rhs_y[g0] = conn.periodicity.data[1] * c;
            }
        }
    }
}

// ---------------------------------------------------------------- smooth.zig:1107-1113
void System::fill(Index iteration) {
    if (iteration > 0) control_function.update(mesh);
    fillBlockInternalPointData();
    fillBlockConnectionData();
}

// ---------------------------------------------------------------- smooth.zig:1115-1143
void System::fillXSpecific() {
    for (const Condition& bc : mesh.boundary_conditions) {
        const int32_t shift = bc.range.firstInternalPointShift(mesh);
        RangeIterator it = bc.range.iterate(mesh);
        Index local;
        while (it.next(local)) {
            Index i, j;
            index_converter.index2d(bc.range.block, local, i, j);
            if (boundary_points.kind[boundary_points.index_converter.bufferIndex(bc.range.block, i, j)] != sliding_circ) continue;
            const Index s = nzStart(index_converter.globalIndex(bc.range.block, local));
            if (shift > 0) {
                lhs_values[s] = 1.0;
                lhs_values[s + 1] = 0.0;
            } else {
                lhs_values[s] = 0.0;
                lhs_values[s + 1] = 1.0;
            }
        }
    }
}

// ---------------------------------------------------------------- smooth.zig:1145-1165
void System::fillYSpecific() {
    for (const Condition& bc : mesh.boundary_conditions) {
        RangeIterator it = bc.range.iterate(mesh);
        Index local;
        while (it.next(local)) {
            Index i, j;
            index_converter.index2d(bc.range.block, local, i, j);
            if (boundary_points.kind[boundary_points.index_converter.bufferIndex(bc.range.block, i, j)] != sliding_circ) continue;
            const Index s = nzStart(index_converter.globalIndex(bc.range.block, local));
            lhs_values[s] = 1.0;
            lhs_values[s + 1] = -1.0;
        }
    }
}

// ---------------------------------------------------------------- BiCGStab.zig:136-153 / GMRES.zig:157-174
void System::seedInitialGuess() {
    Index row = 0;
    for (const Block& block : mesh.blocks)
        for (Index p = 0; p < block.dof(); ++p, ++row) {
            x_new[row] = block.pts[p].data[0];
            y_new[row] = block.pts[p].data[1];
        }
    seeded_initial_guess = true;
}

// ---------------------------------------------------------------- smooth.zig:112-153
Float System::commit(Float* dx2, Float* dy2) {
    Float x_norm_sqr = 0.0, y_norm_sqr = 0.0;
    Index row = 0;
    for (const Block& block : mesh.blocks)
        for (Index p = 0; p < block.dof(); ++p, ++row) {
            const Float dx = block.pts[p].data[0] - x_new[row];
            const Float dy = block.pts[p].data[1] - y_new[row];
            x_norm_sqr += dx * dx;
            y_norm_sqr += dy * dy;
        }
    const Float norm = (x_norm_sqr + y_norm_sqr) * (x_norm_sqr + y_norm_sqr);   // :136
    row = 0;
    for (Block& block : mesh.blocks)
        for (Index p = 0; p < block.dof(); ++p, ++row) block.pts[p] = vinit(x_new[row], y_new[row]);
    if (dx2) *dx2 = x_norm_sqr;
    if (dy2) *dy2 = y_norm_sqr;
    return norm;
}

// ---------------------------------------------------------------- BiCGStab.zig:424-435
void System::matVec(const Float* x, Float* out) const {
    for (Index row = 0; row < dof; ++row) {
        Float sum = 0.0;
        for (int32_t k = lhs_p[row]; k < lhs_p[row + 1]; ++k) sum += lhs_values[k] * x[lhs_i[k]];
        out[row] = sum;
    }
}

}  // namespace orc
