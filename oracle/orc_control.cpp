// ORACLE (test infrastructure) -- Poisson control function (P,Q).
// Follows reference src/core/smoothing/wall_control_function.zig:22-474 (live code only;
// KhamaysehEtAl :476-725 is dead code and not restated).  `white` is hard-coded to blocks
// 0,1 and connection 0 of the O4H template exactly like the reference (:72, :204-213).
#include "orc_refmath.hpp"   // the reference's libm (Zig std.math = musl's acos / atan2), not glibc's
#include "orc_system.hpp"

namespace orc {

namespace {

struct PQ {
    Float p, q;
};

// eq. 6.10 (wall_control_function.zig:100-102, 140-142, 182-184, 257-259)
inline PQ eq610(Float x_xi, Float y_xi, Float x_xi2, Float y_xi2, Float x_eta, Float y_eta, Float x_eta2, Float y_eta2) {
    const Float g11 = x_xi * x_xi + y_xi * y_xi;
    const Float g22 = x_eta * x_eta + y_eta * y_eta;
    const Float p = -(x_xi * x_xi2 + y_xi * y_xi2) / g11 - (x_xi * x_eta2 + y_xi * y_eta2) / g22;
    const Float q = -(x_eta * x_eta2 + y_eta * y_eta2) / g22 - (x_eta * x_xi2 + y_eta * y_xi2) / g11;
    return {p, q};
}

// blend along j: factor = 1 - j/(nj-1)   (:107-111)
inline void blendRow(std::vector<Vec2d>& cf, Index base, Index& local_id, Index size_j, Float p, Float q) {
    cf[base + local_id] = vinit(p, q);
    local_id += 1;
    for (Index j = 1; j < size_j; ++j) {
        const Float factor = 1 - static_cast<Float>(j) / (static_cast<Float>(size_j) - 1);
        cf[base + local_id] = vinit(factor * p, factor * q);
        local_id += 1;
    }
}

void requireO4H(const Mesh& mesh) {
    if (mesh.blocks.size() < 2 || mesh.connections.empty()) throw Error(ORC_E_TOPOLOGY, "white control function needs the O4H layout");
    const Connection& c = mesh.connections[0];
    if (!(c.ranges[0].block == 0 && c.ranges[0].start == 0 && c.ranges[0].side == j_min && c.ranges[1].block == 1 &&
          c.ranges[1].start == 0 && c.ranges[1].side == j_min && !c.has_periodicity))
        throw Error(ORC_E_TOPOLOGY, "white control function: connection 0 must join blocks 0,1 at j_min,start 0 (wall_control_function.zig:212-217)");
    for (int b = 0; b < 2; ++b)
        if (mesh.blocks[b].ni < 3 || mesh.blocks[b].nj < 3) throw Error(ORC_E_SIZE, "white control function: block too small");
}

// wall_control_function.zig:70-280
void whiteInit(std::vector<Vec2d>& cf, const Mesh& mesh) {
    requireO4H(mesh);
    Index block_range_start = 0;
    for (Index b = 0; b < 2; ++b) {
        const Block& block = mesh.blocks[b];
        const Index nj = block.nj;
        const Vec2d* d = block.pts;
        Index local_id = 0;
        {   // corner 0,0: forward differences (:77-112)
            const Float x00 = d[0].data[0], y00 = d[0].data[1];
            const Float x01 = d[1].data[0], y01 = d[1].data[1];
            const Float x02 = d[2].data[0], y02 = d[2].data[1];
            const Float x10 = d[nj].data[0], y10 = d[nj].data[1];
            const Float x20 = d[2 * nj].data[0], y20 = d[2 * nj].data[1];
            const PQ r = eq610(-x00 + x10, -y00 + y10, x00 - 2 * x10 + x20, y00 - 2 * y10 + y20, -x00 + x01, -y00 + y01,
                               x00 - 2 * x01 + x02, y00 - 2 * y01 + y02);
            blendRow(cf, block_range_start, local_id, nj, r.p, r.q);
        }
        for (Index i = 1; i + 1 < block.ni; ++i) {   // edge i_min: central xi, forward eta (:115-153)
            const Float xm = d[local_id - nj].data[0], ym = d[local_id - nj].data[1];
            const Float x0 = d[local_id].data[0], y0 = d[local_id].data[1];
            const Float x1 = d[local_id + 1].data[0], y1 = d[local_id + 1].data[1];
            const Float x2 = d[local_id + 2].data[0], y2 = d[local_id + 2].data[1];
            const Float xp = d[local_id + nj].data[0], yp = d[local_id + nj].data[1];
            const Float x_xi = 0.5 * (xp - xm), y_xi = 0.5 * (yp - ym);
            const Float x_xi2 = xp - 2 * x0 + xm, y_xi2 = yp - 2 * y0 + ym;
            // NOTE operand order of eq610 follows :130-142 (g11 from xi, g22 from eta)
            const PQ r = eq610(x_xi, y_xi, x_xi2, y_xi2, -x0 + x1, -y0 + y1, x0 - 2 * x1 + x2, y0 - 2 * y1 + y2);
            blendRow(cf, block_range_start, local_id, nj, r.p, r.q);
        }
        {   // corner n,0: backward xi, forward eta (:155-194)
            const Float xn0 = d[local_id].data[0], yn0 = d[local_id].data[1];
            const Float xn1 = d[local_id + 1].data[0], yn1 = d[local_id + 1].data[1];
            const Float xn2 = d[local_id + 2].data[0], yn2 = d[local_id + 2].data[1];
            const Float xm1 = d[local_id - nj].data[0], ym1 = d[local_id - nj].data[1];
            const Float xm2 = d[local_id - 2 * nj].data[0], ym2 = d[local_id - 2 * nj].data[1];
            const PQ r = eq610(xn0 - xm1, yn0 - ym1, xn0 - 2 * xm1 + xm2, yn0 - 2 * ym1 + ym2, -xn0 + xn1, -yn0 + yn1,
                               xn0 - 2 * xn1 + xn2, yn0 - 2 * yn1 + yn2);
            blendRow(cf, block_range_start, local_id, nj, r.p, r.q);
        }
        block_range_start += block.dof();
    }
    {   // leading-edge node of block 0 across connection 0 (:203-279)
        const Connection& conn = mesh.connections[0];
        const Vec2d* pd0 = mesh.blocks[conn.ranges[0].block].pts;
        const Vec2d* pd1 = mesh.blocks[conn.ranges[1].block].pts;
        RangeFillMatrixIterator it = RangeFillMatrixIterator::init(conn, mesh);
        Index ids[2] = {0, 0};
        it.next(ids);
        const std::ptrdiff_t p0 = static_cast<std::ptrdiff_t>(ids[0]), p1 = static_cast<std::ptrdiff_t>(ids[1]);
        const Vec2d xij = pd0[p0];
        const Vec2d xip1 = pd0[p0 + it.first_internal_point_shift[0]];
        const Vec2d xim1 = pd1[p1 + it.first_internal_point_shift[1]];
        const Vec2d xjp1 = pd0[p0 + it.in_connection_direction_shift[0]];
        const Vec2d xjp2 = pd0[p0 + 2 * it.in_connection_direction_shift[0]];
        const Float x_xi = 0.5 * (xip1.data[0] - xim1.data[0]), y_xi = 0.5 * (xip1.data[1] - xim1.data[1]);
        const Float x_xi2 = xip1.data[0] - 2 * xij.data[0] + xim1.data[0], y_xi2 = xip1.data[1] - 2 * xij.data[1] + xim1.data[1];
        const Float x_eta = -xij.data[0] + xjp1.data[0], y_eta = -xij.data[1] + xjp1.data[1];
        const Float x_eta2 = xij.data[0] - 2 * xjp1.data[0] + xjp2.data[0], y_eta2 = xij.data[1] - 2 * xjp1.data[1] + xjp2.data[1];
        const PQ r = eq610(x_xi, y_xi, x_xi2, y_xi2, x_eta, y_eta, x_eta2, y_eta2);
        Index local_id = 0;
        blendRow(cf, 0, local_id, mesh.blocks[0].nj, r.p, r.q);
    }
}

// wall_control_function.zig:282-320
void computeUpdate(const White& w, Index& local_id, std::vector<Vec2d>& cf, Float x_xi, Float y_xi, Float x_eta, Float y_eta,
                   Index size_j, Index block_range_start) {
    const Float g11 = x_xi * x_xi + y_xi * y_xi;
    const Float g12 = x_xi * x_eta + y_xi * y_eta;
    const Float g22 = x_eta * x_eta + y_eta * y_eta;
    const Float ds = std::sqrt(g22);
    const Float theta = orc_refmath::acos(g12 / std::sqrt(g11 * g22));
    const Float delta_ds = w.ds_target - ds;
    const Float delta_theta = w.theta_target - theta;
    const Float delta_p = -orc_refmath::atan2(delta_theta, w.theta_target);
    const Float delta_q = orc_refmath::atan2(delta_ds, w.ds_target);
    Float p = cf[block_range_start + local_id].data[0], q = cf[block_range_start + local_id].data[1];
    p += 0.1 * delta_p;
    q += 0.1 * delta_q;
    blendRow(cf, block_range_start, local_id, size_j, p, q);
}

// wall_control_function.zig:322-473
void whiteUpdate(const White& w, std::vector<Vec2d>& cf, const Mesh& mesh) {
    requireO4H(mesh);
    Index block_range_start = 0;
    for (Index b = 0; b < 2; ++b) {
        const Block& block = mesh.blocks[b];
        const Index nj = block.nj;
        const Vec2d* d = block.pts;
        Index local_id = 0;
        {   // :332-348
            const Float x_xi = -d[0].data[0] + d[nj].data[0], y_xi = -d[0].data[1] + d[nj].data[1];
            const Float x_eta = -d[0].data[0] + d[1].data[0], y_eta = -d[0].data[1] + d[1].data[1];
            computeUpdate(w, local_id, cf, x_xi, y_xi, x_eta, y_eta, nj, block_range_start);
        }
        for (Index i = 1; i + 1 < block.ni; ++i) {   // :350-367
            const Float x_xi = 0.5 * (d[local_id + nj].data[0] - d[local_id - nj].data[0]);
            const Float y_xi = 0.5 * (d[local_id + nj].data[1] - d[local_id - nj].data[1]);
            const Float x_eta = -d[local_id].data[0] + d[local_id + 1].data[0];
            const Float y_eta = -d[local_id].data[1] + d[local_id + 1].data[1];
            computeUpdate(w, local_id, cf, x_xi, y_xi, x_eta, y_eta, nj, block_range_start);
        }
        {   // :369-385
            const Float x_xi = d[local_id].data[0] - d[local_id - nj].data[0], y_xi = d[local_id].data[1] - d[local_id - nj].data[1];
            const Float x_eta = -d[local_id].data[0] + d[local_id + 1].data[0], y_eta = -d[local_id].data[1] + d[local_id + 1].data[1];
            computeUpdate(w, local_id, cf, x_xi, y_xi, x_eta, y_eta, nj, block_range_start);
        }
        block_range_start += block.dof();
    }
    {   // leading-edge node (:393-472); note the NEGATED xi difference (:428-431)
        const Connection& conn = mesh.connections[0];
        const Vec2d* pd0 = mesh.blocks[conn.ranges[0].block].pts;
        const Vec2d* pd1 = mesh.blocks[conn.ranges[1].block].pts;
        RangeFillMatrixIterator it = RangeFillMatrixIterator::init(conn, mesh);
        Index ids[2] = {0, 0};
        it.next(ids);
        const std::ptrdiff_t p0 = static_cast<std::ptrdiff_t>(ids[0]), p1 = static_cast<std::ptrdiff_t>(ids[1]);
        const Vec2d xij = pd0[p0];
        const Vec2d xip1 = pd0[p0 + it.first_internal_point_shift[0]];
        const Vec2d xim1 = pd1[p1 + it.first_internal_point_shift[1]];
        const Vec2d xjp1 = pd0[p0 + it.in_connection_direction_shift[0]];
        const Float x_xi = -0.5 * (xip1.data[0] - xim1.data[0]), y_xi = -0.5 * (xip1.data[1] - xim1.data[1]);
        const Float x_eta = -xij.data[0] + xjp1.data[0], y_eta = -xij.data[1] + xjp1.data[1];
        const Float g11 = x_xi * x_xi + y_xi * y_xi;
        const Float g12 = x_xi * x_eta + y_xi * y_eta;
        const Float g22 = x_eta * x_eta + y_eta * y_eta;
        const Float ds = std::sqrt(g22);
        const Float theta = orc_refmath::acos(g12 / std::sqrt(g11 * g22));
        const Float delta_p = -orc_refmath::atan2(w.theta_target - theta, w.theta_target);
        const Float delta_q = orc_refmath::atan2(w.ds_target - ds, w.ds_target);
        Float p = cf[0].data[0], q = cf[0].data[1];
        p += 0.1 * delta_p;
        q += 0.1 * delta_q;
        Index local_id = 0;
        blendRow(cf, 0, local_id, mesh.blocks[0].nj, p, q);
    }
}

}  // namespace

// wall_control_function.zig:27-42
void ControlFunction::init(Index dof, const Mesh& mesh, int algo, White w) {
    data.assign(dof, vinit(0, 0));
    algorithm = algo;
    white = w;
    if (algo == ORC_CF_WHITE) whiteInit(data, mesh);
}

// wall_control_function.zig:48-53
void ControlFunction::update(const Mesh& mesh) {
    if (algorithm == ORC_CF_WHITE) whiteUpdate(white, data, mesh);
}

}  // namespace orc


extern "C" void orc_ref_white_math(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2) {
    for (uint64_t i = 0; i < n; ++i) {
        out_acos[i] = orc_refmath::acos(x[i]);
        out_atan2[i] = orc_refmath::atan2(y[i], x[i]);
    }
}
