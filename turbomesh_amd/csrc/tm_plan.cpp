// Host-side planning: row classification and the perimeter-row table.
//
// Semantics follow the reference's system assembly (src/core/smoothing/smooth.zig):
//   - row kinds and their order-dependent assignment      smooth.zig:1234-1332
//   - junction ("laplacian") points                       smooth.zig:1340-1514
//   - column pattern of perimeter rows                    smooth.zig:421-458, 518-778
//   - static coefficients / right-hand sides              smooth.zig:780-921, 1115-1165
// but produce a compact per-row table instead of a global CSR: interior rows are never
// materialised (K2 is matrix-free), `smoothed` interface rows keep only their column ids, the
// stencil slot feeding each column and the four metric neighbours (coefficients are
// recomputed on the device from the frozen coordinates every application).
#include "tm_plan.hpp"
#include "../../include/tm_hip.h"
#include <cstdlib>
#include <algorithm>
#include <unordered_map>

namespace tmh {

namespace {

struct Junction {
    std::vector<int64_t> ids;       // overlapping global ids (sorted ascending at the end)
    std::vector<double> per;        // periodicity per overlapping id (x,y interleaved)
    std::vector<int64_t> stencil;   // sorted stencil ids, includes ids[0]
    double rhs[2] = {0.0, 0.0};
};

// stencil slots, smooth.zig:175-185
enum { S_I_J = 0, S_IP1_J, S_IM1_J, S_I_JP1, S_I_JM1, S_IP1_JP1, S_IP1_JM1, S_IM1_JP1, S_IM1_JM1 };

struct Builder {
    const Topology& t;
    std::vector<int64_t> perim_start;   // first row index of each block in rows[]
    std::vector<PlanRow> rows;

    explicit Builder(const Topology& topo) : t(topo) {}

    int64_t perim_count(int64_t b) const { return 2 * (t.nj[b] + t.ni[b] - 2); }
    void block_of(int64_t gid, int64_t& b, int64_t& local) const {
        int64_t k = t.nblocks() - 1;
        while (gid < t.start[k]) --k;
        b = k;
        local = gid - t.start[k];
    }
    // perimeter numbering of boundary.zig:248-285
    int64_t perim_index(int64_t b, int64_t i, int64_t j) const {
        const int64_t ni = t.ni[b], nj = t.nj[b];
        if (i == 0) return j;
        if (i == ni - 1) return nj + 2 * (ni - 2) + j;
        if (j == 0) return nj + (i - 1) * 2;
        if (j == nj - 1) return nj - 1 + i * 2;
        throw PlanError(TM_E_TOPOLOGY, "NotBoundaryIndex: node is not on the block perimeter");
    }
    PlanRow& row_of(int64_t gid) {
        int64_t b, local;
        block_of(gid, b, local);
        const int64_t i = local / t.nj[b], j = local - i * t.nj[b];
        return rows[perim_start[b] + perim_index(b, i, j)];
    }
    // boundary.zig:28-61 Range.iterate -> block-local flat ids
    std::vector<int64_t> range_points(const TopoRange& r) const {
        const int64_t ni = t.ni[r.block], nj = t.nj[r.block];
        int64_t idx = 0, inc = 0;
        switch (r.side) {
            case SIDE_I_MIN: idx = r.start * nj; inc = nj; break;
            case SIDE_I_MAX: idx = r.start * nj + nj - 1; inc = nj; break;
            case SIDE_J_MIN: idx = r.start; inc = 1; break;
            case SIDE_J_MAX: idx = (ni - 1) * nj + r.start; inc = 1; break;
        }
        const int64_t n = (r.start > r.end ? r.start - r.end : r.end - r.start) + 1;
        if (r.start > r.end) inc = -inc;
        std::vector<int64_t> out(n);
        for (int64_t k = 0; k < n; ++k) out[k] = idx + k * inc;
        return out;
    }
    int64_t first_internal_shift(const TopoRange& r) const {   // boundary.zig:78-97
        switch (r.side) {
            case SIDE_I_MIN: return 1;
            case SIDE_I_MAX: return -1;
            case SIDE_J_MIN: return t.nj[r.block];
            default: return -t.nj[r.block];
        }
    }

    std::vector<Junction> find_junctions() const;
    void build();
};

void append_unique(Junction& jn, int64_t id, const double per[2]) {   // smooth.zig:1516-1522
    for (int64_t v : jn.ids)
        if (v == id) return;
    if (jn.ids.size() >= 4) throw PlanError(TM_E_OVERFLOW, "junction point with more than 4 overlapping points");
    jn.ids.push_back(id);
    jn.per.push_back(per[0]);
    jn.per.push_back(per[1]);
}

// smooth.zig:1340-1514.  A mesh without connections has no junctions (the reference cannot
// run such a mesh at all: usize underflow at smooth.zig:1364).
std::vector<Junction> Builder::find_junctions() const {
    std::vector<Junction> out;
    const size_t nc = t.conns.size();
    if (nc == 0) return out;
    std::vector<int64_t> ep(nc * 4);
    for (size_t c = 0; c < nc; ++c) {
        const auto p0 = range_points(t.conns[c].r[0]);
        const auto p1 = range_points(t.conns[c].r[1]);
        ep[c * 4 + 0] = t.start[t.conns[c].r[0].block] + p0.front();
        ep[c * 4 + 1] = t.start[t.conns[c].r[1].block] + p1.front();
        ep[c * 4 + 2] = t.start[t.conns[c].r[0].block] + p0.back();
        ep[c * 4 + 3] = t.start[t.conns[c].r[1].block] + p1.back();
    }
    const double zero[2] = {0.0, 0.0};
    auto per_of = [&](size_t conn) -> const double* { return t.conns[conn].periodic ? t.conns[conn].per : zero; };
    for (size_t e = 0; e + 1 < ep.size(); ++e) {
        for (size_t chk = e + 1; chk < ep.size(); ++chk) {
            if (ep[chk] != ep[e]) continue;
            bool found = false;
            for (auto& jn : out) {
                const size_t n0 = jn.ids.size();
                for (size_t k = 0; k < n0; ++k)
                    if (jn.ids[k] == ep[e]) {
                        found = true;
                        const size_t add = (chk % 2 == 0) ? chk + 1 : chk - 1;   // the id paired with the match
                        append_unique(jn, ep[add], per_of(add / 4));
                    }
            }
            if (found) continue;
            const size_t p0 = e / 2, p1 = chk / 2;   // end-point pairs
            if (p0 == p1) throw PlanError(TM_E_TOPOLOGY, "connection joins a point with itself");
            Junction jn;
            jn.ids.push_back(ep[p0 * 2]);
            jn.per.insert(jn.per.end(), {0.0, 0.0});
            append_unique(jn, ep[p0 * 2 + 1], per_of(p0 / 2));
            if (jn.ids.size() != 2) throw PlanError(TM_E_TOPOLOGY, "connection joins a point with itself");
            append_unique(jn, ep[p1 * 2], per_of(p1 / 2));
            append_unique(jn, ep[p1 * 2 + 1], per_of(p1 / 2));
            out.push_back(jn);
        }
    }
    for (auto& jn : out) {   // sort overlapping ids (with their periodicity), smooth.zig:1442-1448
        std::vector<size_t> order(jn.ids.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jn.ids[a] < jn.ids[b]; });
        Junction s = jn;
        for (size_t k = 0; k < order.size(); ++k) {
            s.ids[k] = jn.ids[order[k]];
            s.per[2 * k] = jn.per[2 * order[k]];
            s.per[2 * k + 1] = jn.per[2 * order[k] + 1];
        }
        jn = s;
    }
    std::stable_sort(out.begin(), out.end(), [](const Junction& a, const Junction& b) { return a.ids[0] < b.ids[0]; });
    for (auto& jn : out) {   // stencil: the junction + first-interior diagonal neighbours, smooth.zig:1458-1511
        jn.stencil.push_back(jn.ids[0]);
        for (size_t k = 0; k < jn.ids.size(); ++k) {
            int64_t b, local;
            block_of(jn.ids[k], b, local);
            const int64_t ni = t.ni[b], nj = t.nj[b];
            const int64_t i = local / nj, j = local - i * nj;
            int64_t nb[2][2];
            int n = 0;
            auto push = [&](int64_t a, int64_t c) { nb[n][0] = a; nb[n][1] = c; ++n; };
            if (i == 0) {
                if (j == 0) push(1, 1);
                else if (j == nj - 1) push(1, nj - 2);
                else { push(1, j - 1); push(1, j + 1); }
            } else if (i == ni - 1) {
                if (j == 0) push(ni - 2, 1);
                else if (j == nj - 1) push(ni - 2, nj - 2);
                else { push(ni - 2, j - 1); push(ni - 2, j + 1); }
            } else if (j == 0) {
                push(i - 1, 1);
                push(i + 1, 1);
            } else if (j == nj - 1) {
                push(i - 1, j - 1);
                push(i + 1, j - 1);
            } else {
                throw PlanError(TM_E_TOPOLOGY, "connection end point is not on the block perimeter");
            }
            for (int q = 0; q < n; ++q) {
                if (jn.stencil.size() >= 6) throw PlanError(TM_E_OVERFLOW, "junction stencil with more than 6 points");
                jn.stencil.push_back(t.start[b] + nb[q][0] * nj + nb[q][1]);
                jn.rhs[0] += jn.per[2 * k];
                jn.rhs[1] += jn.per[2 * k + 1];
            }
        }
        std::sort(jn.stencil.begin(), jn.stencil.end());
    }
    return out;
}

// stencil column positions of a `smoothed` row, smooth.zig:518-616
void stencil_positions(const TopoConn& c, const ConnShifts& s, int pos[9]) {
    if (c.r[0].block == c.r[1].block) {
        if (!(c.r[0].side == SIDE_I_MIN && c.r[1].side == SIDE_I_MAX))
            throw PlanError(TM_E_TOPOLOGY, "a connection inside one block must run i_min -> i_max (smooth.zig:522-559)");
        const int up[9] = {1, 4, 7, 0, 3, 6, 2, 5, 8}, down[9] = {7, 4, 1, 6, 3, 0, 8, 5, 2};
        const bool fwd = s.direction[0] > 0;
        if (fwd != (s.direction[1] > 0)) throw PlanError(TM_E_TOPOLOGY, "self connection with opposite directions");
        for (int k = 0; k < 9; ++k) pos[k] = fwd ? up[k] : down[k];
        return;
    }
    if (!(c.r[0].block < c.r[1].block)) throw PlanError(TM_E_TOPOLOGY, "connection needs ranges[0].block < ranges[1].block (smooth.zig:562)");
    const int d0 = s.direction[0] > 0 ? 1 : -1, d1 = s.direction[1] > 0 ? 1 : -1;
    switch (c.r[0].side) {
        case SIDE_I_MIN: pos[0] = 3 - 2 * d0; pos[1] = 3; pos[2] = 3 + 2 * d0; pos[3] = 2 - 2 * d0; pos[4] = 2; pos[5] = 2 + 2 * d0; break;
        case SIDE_I_MAX: pos[0] = 2 - 2 * d0; pos[1] = 2; pos[2] = 2 + 2 * d0; pos[3] = 3 - 2 * d0; pos[4] = 3; pos[5] = 3 + 2 * d0; break;
        case SIDE_J_MIN: pos[0] = 4 - d0; pos[1] = 4; pos[2] = 4 + d0; pos[3] = 1 - d0; pos[4] = 1; pos[5] = 1 + d0; break;
        default: pos[0] = 1 - d0; pos[1] = 1; pos[2] = 1 + d0; pos[3] = 4 - d0; pos[4] = 4; pos[5] = 4 + d0; break;
    }
    pos[6] = 7 - d1;
    pos[7] = 7;
    pos[8] = 7 + d1;
}

void Builder::build() {
    const int64_t nb = t.nblocks();
    perim_start.resize(nb);
    int64_t total = 0;
    for (int64_t b = 0; b < nb; ++b) {
        perim_start[b] = total;
        total += perim_count(b);
    }
    rows.assign(total, PlanRow{});
    // rows in perimeter order == ascending global id
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t ni = t.ni[b], nj = t.nj[b];
        auto init = [&](int64_t i, int64_t j) {
            PlanRow& r = rows[perim_start[b] + perim_index(b, i, j)];
            r.gid = t.start[b] + i * nj + j;
            r.kind = KIND_FIXED;   // smooth.zig:1243
            r.ncols = 0;
            r.self = 0;
            for (int k = 0; k < 9; ++k) {
                r.col[k] = -1;
                r.cx[k] = r.cy[k] = 0.0;
                r.slot[k] = 0;
            }
            for (int k = 0; k < 4; ++k) r.metric[k] = -1;
            r.per[0] = r.per[1] = 0.0;
            r.flags = 0;
            r.rhs[0] = r.rhs[1] = 0.0;
            r.rhs_coord = 0;
        };
        for (int64_t j = 0; j < nj; ++j) init(0, j);
        for (int64_t i = 1; i + 1 < ni; ++i) {
            init(i, 0);
            init(i, nj - 1);
        }
        for (int64_t j = 0; j < nj; ++j) init(ni - 1, j);
    }

    // ---- kinds, in the reference's order (later steps overwrite earlier ones) ----
    const std::vector<Junction> junctions = find_junctions();
    for (const auto& jn : junctions) {   // smooth.zig:1246-1263
        row_of(jn.ids[0]).kind = KIND_JUNCTION;
        for (size_t k = 1; k < jn.ids.size(); ++k) row_of(jn.ids[k]).kind = KIND_CONNECTED;
    }
    for (const auto& bc : t.bcs) {   // smooth.zig:1265-1277
        if (bc.kind == TM_BC_WALL) continue;
        for (int64_t p : range_points(bc.range)) row_of(t.start[bc.range.block] + p).kind = KIND_SLIDING;
    }
    for (const auto& c : t.conns) {   // smooth.zig:1280-1329
        const auto p0 = range_points(c.r[0]), p1 = range_points(c.r[1]);
        const int64_t s0 = t.start[c.r[0].block], s1 = t.start[c.r[1].block];
        auto endpoint = [&](size_t k) {
            const int8_t k0 = row_of(s0 + p0[k]).kind;
            if (k0 == KIND_FIXED || k0 == KIND_SLIDING) row_of(s1 + p1[k]).kind = KIND_CONNECTED;
        };
        endpoint(0);
        for (size_t k = 1; k + 1 < p0.size(); ++k) {
            row_of(s0 + p0[k]).kind = KIND_SMOOTHED;
            row_of(s1 + p1[k]).kind = KIND_CONNECTED;
        }
        endpoint(p0.size() - 1);
    }

    // ---- column pattern + static coefficients ----
    for (PlanRow& r : rows) {   // defaults per kind, smooth.zig:421-458, 780-865
        switch (r.kind) {
            case KIND_FIXED:
                r.ncols = 1;
                r.col[0] = r.gid;
                r.cx[0] = r.cy[0] = 1.0;
                r.rhs_coord = 3;
                break;
            case KIND_SMOOTHED: r.ncols = 9; break;
            case KIND_CONNECTED:
                r.ncols = 2;
                r.cx[0] = r.cy[0] = 1.0;
                r.cx[1] = r.cy[1] = -1.0;
                r.self = 1;
                break;
            case KIND_SLIDING:
                r.ncols = 2;
                r.rhs_coord = 1;   // x-system: x_self = x_boundary; y-system rhs 0
                break;
            default: break;   // junction rows are filled below
        }
    }
    for (const auto& jn : junctions) {   // smooth.zig:444-448, 813-836, 738-747, 917-920
        PlanRow& r = row_of(jn.ids[0]);
        r.ncols = static_cast<int8_t>(jn.stencil.size());
        for (size_t k = 0; k < jn.stencil.size(); ++k) {
            r.col[k] = jn.stencil[k];
            r.cx[k] = r.cy[k] = 1.0;
            if (jn.stencil[k] == jn.ids[0]) r.self = static_cast<int8_t>(k);
        }
        r.cx[r.self] = r.cy[r.self] = -static_cast<double>(jn.stencil.size()) + 1;
        for (size_t k = 1; k < jn.ids.size(); ++k) {
            PlanRow& o = row_of(jn.ids[k]);
            if (o.kind != KIND_CONNECTED)
                throw PlanError(TM_E_TOPOLOGY, "junction partner row is not a `connected` row (kinds were overwritten)");
            o.col[0] = jn.ids[0];
            o.col[1] = jn.ids[k];
        }
    }
    for (const auto& c : t.conns) {   // smooth.zig:618-721
        const ConnShifts s = conn_shifts(t, c);
        if (!(s.count > 2 && s.count - 2 > 3))
            throw PlanError(TM_E_TOPOLOGY, "a connection needs more than 3 interior points (smooth.zig:631)");
        int pos[9];
        stencil_positions(c, s, pos);
        const int64_t s0 = t.start[c.r[0].block], s1 = t.start[c.r[1].block];
        auto endpoint = [&](int64_t l0, int64_t l1) {   // smooth.zig:695-721
            const int64_t g0 = s0 + l0, g1 = s1 + l1;
            switch (row_of(g0).kind) {
                case KIND_FIXED:
                case KIND_SLIDING: {
                    PlanRow& o = row_of(g1);
                    if (o.kind != KIND_CONNECTED) throw PlanError(TM_E_TOPOLOGY, "connection end point: side 1 is not a `connected` row");
                    if (!(g0 < g1)) throw PlanError(TM_E_TOPOLOGY, "connection end point ids must ascend from side 0 to side 1");
                    o.col[0] = g0;
                    o.col[1] = g1;
                    break;
                }
                case KIND_JUNCTION:
                case KIND_CONNECTED: break;
                default: throw PlanError(TM_E_TOPOLOGY, "connection end point on side 0 is a `smoothed` point (smooth.zig:719)");
            }
        };
        endpoint(s.position[0], s.position[1]);
        for (int64_t k = 1; k + 1 < s.count; ++k) {
            const int64_t l0 = s.position[0] + k * s.direction[0], l1 = s.position[1] + k * s.direction[1];
            const int64_t g0 = s0 + l0, g1 = s1 + l1;
            PlanRow& o = row_of(g1);
            PlanRow& r = row_of(g0);
            if (o.kind != KIND_CONNECTED || r.kind != KIND_SMOOTHED)
                throw PlanError(TM_E_TOPOLOGY, "overlapping connections: an interface point belongs to two connection interiors");
            if (!(g0 < g1)) throw PlanError(TM_E_TOPOLOGY, "connected point ids must ascend from side 0 to side 1 (smooth.zig:651)");
            o.col[0] = g0;
            o.col[1] = g1;
            const int64_t d0 = s.direction[0], d1 = s.direction[1], f0 = s.first_internal[0], f1 = s.first_internal[1];
            const int64_t ids[9] = {g0 - d0 + f0, g0 + f0, g0 + d0 + f0, g0 - d0, g0, g0 + d0, g1 - d1 + f1, g1 + f1, g1 + d1 + f1};
            const int8_t slots[9] = {S_IM1_JM1, S_I_JM1, S_IP1_JM1, S_IM1_J, S_I_J, S_IP1_J, S_IM1_JP1, S_I_JP1, S_IP1_JP1};
            for (int q = 0; q < 9; ++q) {
                r.col[pos[q]] = ids[q];
                r.slot[pos[q]] = slots[q];
            }
            for (int q = 0; q < 8; ++q)
                if (!(r.col[q] < r.col[q + 1])) throw PlanError(TM_E_TOPOLOGY, "interface stencil columns do not ascend (smooth.zig:679-687)");
            r.self = static_cast<int8_t>(pos[4]);
            r.metric[0] = g0 - d0;
            r.metric[1] = g0 + d0;
            r.metric[2] = g0 + f0;
            r.metric[3] = g1 + f1;
            r.flags = c.periodic ? 1 : 2;   // periodic rows pass (P,Q), the others (Q,P): smooth.zig:1040-1041, 1082-1083
            r.per[0] = c.periodic ? c.per[0] : 0.0;
            r.per[1] = c.periodic ? c.per[1] : 0.0;
        }
        endpoint(s.position[0] + (s.count - 1) * s.direction[0], s.position[1] + (s.count - 1) * s.direction[1]);
    }
    for (const auto& bc : t.bcs) {   // smooth.zig:751-777, 1115-1165
        if (bc.kind == TM_BC_WALL) throw PlanError(TM_E_TOPOLOGY, "wall conditions must not be listed (smooth.zig:775: unreachable)");
        const int64_t shift = first_internal_shift(bc.range);
        for (int64_t p : range_points(bc.range)) {
            const int64_t g = t.start[bc.range.block] + p;
            PlanRow& r = row_of(g);
            if (r.kind != KIND_SLIDING) continue;
            if (shift > 0) {
                r.col[0] = g;
                r.col[1] = g + shift;
                r.self = 0;
                r.cx[0] = 1.0; r.cx[1] = 0.0;
            } else {
                r.col[0] = g + shift;
                r.col[1] = g;
                r.self = 1;
                r.cx[0] = 0.0; r.cx[1] = 1.0;
            }
            r.cy[0] = 1.0;
            r.cy[1] = -1.0;
        }
    }
    for (const auto& c : t.conns) {   // periodic right-hand sides of the slaved side, smooth.zig:904-915
        if (!c.periodic) continue;
        for (int64_t p : range_points(c.r[1])) {
            PlanRow& r = row_of(t.start[c.r[1].block] + p);
            r.rhs[0] = -c.per[0];
            r.rhs[1] = -c.per[1];
            r.rhs_coord = 0;
        }
    }
    for (const auto& jn : junctions) {   // smooth.zig:917-920
        PlanRow& r = row_of(jn.ids[0]);
        r.rhs[0] = jn.rhs[0];
        r.rhs[1] = jn.rhs[1];
        r.rhs_coord = 0;
    }
    for (const PlanRow& r : rows)
        for (int k = 0; k < r.ncols; ++k)
            if (r.col[k] < 0 || r.col[k] >= t.dof) throw PlanError(TM_E_TOPOLOGY, "perimeter row " + std::to_string(r.gid) + " was never wired to its neighbours");
}

}  // namespace

void Topology::finalize() {
    const int64_t nb = nblocks();
    if (nb == 0) throw PlanError(TM_E_ARG, "mesh has no blocks");
    start.resize(nb);
    dof = 0;
    for (int64_t b = 0; b < nb; ++b) {
        if (ni[b] < 3 || nj[b] < 3) throw PlanError(TM_E_SIZE, "every block needs at least 3 x 3 nodes");
        start[b] = dof;
        dof += ni[b] * nj[b];
    }
    if (dof >= (int64_t{1} << 31)) throw PlanError(TM_E_SIZE, "mesh exceeds 2^31 nodes");
    auto check = [&](const TopoRange& r) {
        if (r.block < 0 || r.block >= nb || r.side > 3) throw PlanError(TM_E_TOPOLOGY, "range names a missing block or side");
        const int64_t lim = (r.side == SIDE_I_MIN || r.side == SIDE_I_MAX) ? ni[r.block] : nj[r.block];
        if (r.start < 0 || r.end < 0 || r.start >= lim || r.end >= lim) throw PlanError(TM_E_TOPOLOGY, "range exceeds its block side");
    };
    for (const auto& c : conns) {
        check(c.r[0]);
        check(c.r[1]);
        const int64_t l0 = std::abs(c.r[0].end - c.r[0].start), l1 = std::abs(c.r[1].end - c.r[1].start);
        if (l0 != l1) throw PlanError(TM_E_MISMATCH, "the two ranges of a connection differ in length");
    }
    for (const auto& bc : bcs) {
        check(bc.range);
        if (bc.kind > 2) throw PlanError(TM_E_TOPOLOGY, "unknown boundary condition kind");
    }
}

ConnShifts conn_shifts(const Topology& t, const TopoConn& c) {   // smooth.zig:1556-1598
    ConnShifts s{};
    for (int k = 0; k < 2; ++k) {
        const TopoRange& r = c.r[k];
        const int64_t ni = t.ni[r.block], nj = t.nj[r.block];
        switch (r.side) {
            case SIDE_I_MIN: s.first_internal[k] = 1; s.direction[k] = nj; s.position[k] = r.start * nj; break;
            case SIDE_I_MAX: s.first_internal[k] = -1; s.direction[k] = nj; s.position[k] = r.start * nj + nj - 1; break;
            case SIDE_J_MIN: s.first_internal[k] = nj; s.direction[k] = 1; s.position[k] = r.start; break;
            default: s.first_internal[k] = -nj; s.direction[k] = 1; s.position[k] = (ni - 1) * nj + r.start; break;
        }
        if (r.start > r.end) {
            s.direction[k] = -s.direction[k];
            s.count = r.start - r.end + 1;
        } else {
            s.count = r.end - r.start + 1;
        }
    }
    return s;
}

std::vector<PlanRow> build_rows(const Topology& t) {
    Builder b(t);
    b.build();
    return std::move(b.rows);
}

int64_t LocalPlan::to_local(int64_t gid) const {
    // owned?
    int64_t b = topo->nblocks() - 1;
    while (gid < topo->start[b]) --b;
    const auto it = std::lower_bound(owned_blocks.begin(), owned_blocks.end(), b);
    if (it != owned_blocks.end() && *it == b) return local_start[it - owned_blocks.begin()] + (gid - topo->start[b]);
    const auto g = ghost_index.find(gid);
    return g == ghost_index.end() ? -1 : n_owned + g->second;
}

// Sweep triples across ranks: one exchange of a depth-3 halo per three sweeps instead of one of a depth-2 halo per two.  Round 3 found
// them slower than pairs below ~2^19 nodes per block (three level launches per triple on the chain: 512^2 10.0 against 9.1 us per sweep);
// since the three level passes run as ONE launch (k_edge_levels3) triples win at every size -- rank 1 of 3, null transport, us per sweep
// triples / pairs (tools/dev/triples_threshold.sh): 256^2 4.2 / 7.6, 512^2 4.5 / 8.0, 1024^2 5.4 / 10.2; with 10 us of exchange on the
// chain 7.5 / 12.8, 7.4 / 13.2, 8.1 / 14.6 -- so every block of at least 16 x 16 nodes takes them.  A pure function of the topology
// (TM_TRIPLES_MIN_NODES overrides the threshold: tests, A/B runs; negative = never), identical on every rank.
bool triple_halo_for(const Topology& t, int nranks) {
    if (nranks < 2) return false;
    int64_t min_nodes = 0;
    if (const char* e = std::getenv("TM_TRIPLES_MIN_NODES")) min_nodes = std::atoll(e);
    if (min_nodes < 0) return false;
    for (int64_t b = 0; b < t.nblocks(); ++b)
        if (t.ni[b] < 16 || t.nj[b] < 16 || t.ni[b] * t.nj[b] < min_nodes) return false;
    return true;
}

LocalPlan build_local_plan(const Topology& t, const std::vector<PlanRow>& all_rows, const std::vector<int32_t>& owner, int rank,
                           int nranks, bool allow_triples) {
    LocalPlan lp;
    lp.rank = rank;
    lp.nranks = nranks;
    lp.topo = &t;
    const int64_t nb = t.nblocks();
    if (static_cast<int64_t>(owner.size()) != nb) throw PlanError(TM_E_ARG, "owner table must have one entry per block");
    for (int64_t b = 0; b < nb; ++b) {
        if (owner[b] < 0 || owner[b] >= nranks) throw PlanError(TM_E_ARG, "owner table names a missing rank");
        if (owner[b] == rank) {
            lp.owned_blocks.push_back(b);
            lp.local_start.push_back(lp.n_owned);
            lp.n_owned += t.ni[b] * t.nj[b];
        }
    }
    auto owner_of_gid = [&](int64_t gid) {
        int64_t b = nb - 1;
        while (gid < t.start[b]) --b;
        return owner[b];
    };
    // ghost sets of EVERY rank (every rank computes the same tables, so no setup communication is needed)
    std::vector<std::vector<std::pair<int32_t, int64_t>>> need(nranks);   // need[r] = (owner, gid) pairs rank r reads remotely
    for (const PlanRow& r : all_rows) {
        const int32_t ro = owner_of_gid(r.gid);
        auto touch = [&](int64_t gid) {
            const int32_t o = owner_of_gid(gid);
            if (o != ro) need[ro].emplace_back(o, gid);
        };
        for (int k = 0; k < r.ncols; ++k) touch(r.col[k]);
        if (r.kind == KIND_SMOOTHED)
            for (int k = 0; k < 4; ++k) touch(r.metric[k]);
        if (ro == rank) lp.rows.push_back(r);
    }
    for (auto& v : need) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    // depth 2: the definition of every depth-1 ghost row, and the remote rows IT reads
    auto row_def = [&](int64_t gid) {
        const auto it = std::lower_bound(all_rows.begin(), all_rows.end(), gid, [](const PlanRow& a, int64_t g) { return a.gid < g; });
        if (it != all_rows.end() && it->gid == gid) return *it;
        int64_t b = nb - 1;
        while (gid < t.start[b]) --b;
        const int64_t nj = t.nj[b];
        PlanRow r{};
        r.gid = gid;
        r.kind = KIND_INTERIOR;
        r.ncols = 9;
        r.self = 4;
        int q = 0;
        for (int64_t di = -1; di <= 1; ++di)
            for (int64_t dj = -1; dj <= 1; ++dj) r.col[q++] = gid + di * nj + dj;
        return r;
    };
    std::vector<std::vector<std::pair<int32_t, int64_t>>> need2 = need;
    for (int r = 0; r < nranks; ++r) {
        for (const auto& pr : need[r]) {
            const PlanRow def = row_def(pr.second);
            auto touch = [&](int64_t gid) {
                const int32_t o = owner_of_gid(gid);
                if (o != r) need2[r].emplace_back(o, gid);
            };
            for (int k = 0; k < def.ncols; ++k) touch(def.col[k]);
            if (def.kind == KIND_SMOOTHED)
                for (int k = 0; k < 4; ++k) touch(def.metric[k]);
            if (r == rank) lp.ghost_rows.push_back(def);
        }
        std::sort(need2[r].begin(), need2[r].end());
        need2[r].erase(std::unique(need2[r].begin(), need2[r].end()), need2[r].end());
    }
    need.swap(need2);
    // depth 3 (sweep TRIPLES on large coupled blocks, Smoother::relax_triples_coupled): the rows of the depth-2 set are evaluated one
    // level further down (ghost_rows2: their definitions), and the remote rows THOSE read travel too.  Decided from the topology -- and
    // from whether the caller's handles will run triples at all (allow_triples: a pure function of the solver options, identical on every
    // rank; a Krylov / White / single-sweep handle exchanges the depth-2 set only) -- so every rank and the transport's tables, which are
    // built from this function as well (tm_rccl_hooks / tm_rccl_hooks_for), agree on the exchange lists.
    lp.triple_halo = allow_triples && triple_halo_for(t, nranks);
    if (lp.triple_halo) {
        std::vector<std::vector<std::pair<int32_t, int64_t>>> need3 = need;
        for (int r = 0; r < nranks; ++r) {
            for (const auto& pr : need[r]) {
                const PlanRow def = row_def(pr.second);
                auto touch = [&](int64_t gid) {
                    const int32_t o = owner_of_gid(gid);
                    if (o != r) need3[r].emplace_back(o, gid);
                };
                for (int k = 0; k < def.ncols; ++k) touch(def.col[k]);
                if (def.kind == KIND_SMOOTHED)
                    for (int k = 0; k < 4; ++k) touch(def.metric[k]);
                if (r == rank) lp.ghost_rows2.push_back(def);
            }
            std::sort(need3[r].begin(), need3[r].end());
            need3[r].erase(std::unique(need3[r].begin(), need3[r].end()), need3[r].end());
        }
        need.swap(need3);
    }
    for (const auto& pr : need[rank]) {
        lp.ghost_index[pr.second] = static_cast<int64_t>(lp.ghost_gid.size());
        lp.ghost_gid.push_back(pr.second);
    }
    // peers: ranks I receive from or send to
    std::vector<std::vector<int64_t>> send_to(nranks), recv_from(nranks);
    for (const auto& pr : need[rank]) recv_from[pr.first].push_back(pr.second);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        for (const auto& pr : need[r])
            if (pr.first == rank) send_to[r].push_back(pr.second);
    }
    int64_t soff = 0, roff = 0;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank || (send_to[r].empty() && recv_from[r].empty())) continue;
        lp.peer_rank.push_back(r);
        lp.send_off.push_back(soff);
        lp.send_cnt.push_back(static_cast<int64_t>(send_to[r].size()));
        lp.recv_off.push_back(roff);
        lp.recv_cnt.push_back(static_cast<int64_t>(recv_from[r].size()));
        for (int64_t gid : send_to[r]) {
            int64_t b = nb - 1;
            while (gid < t.start[b]) --b;
            const auto it = std::lower_bound(lp.owned_blocks.begin(), lp.owned_blocks.end(), b);
            lp.send_ids.push_back(static_cast<int32_t>(lp.local_start[it - lp.owned_blocks.begin()] + (gid - t.start[b])));
        }
        soff += static_cast<int64_t>(send_to[r].size());
        roff += static_cast<int64_t>(recv_from[r].size());
    }
    lp.direct_send = !lp.peer_rank.empty();
    for (size_t k = 0; k < lp.peer_rank.size(); ++k) {
        const int64_t o = lp.send_off[k], c = lp.send_cnt[k];
        lp.send_first.push_back(c ? lp.send_ids[o] : 0);
        for (int64_t q = 1; q < c; ++q)
            if (lp.send_ids[o + q] != lp.send_ids[o] + q) lp.direct_send = false;
    }
    if (lp.n_owned + static_cast<int64_t>(lp.ghost_gid.size()) >= (int64_t{1} << 31)) throw PlanError(TM_E_SIZE, "rank-local vector exceeds 2^31 rows");
    return lp;
}

}  // namespace tmh
