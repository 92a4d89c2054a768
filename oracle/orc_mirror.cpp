// ORACLE (test infrastructure) -- MIRROR of the device's interior-row evaluation.
//
// The HIP kernel K2 (turbomesh_amd/csrc/tm_kernels.hip, winslow_row) evaluates an interior row of
// the reference's system (coefficients smooth.zig:171-216, row smooth.zig:923-992) in an
// algebraically factored form with explicit fused multiply-adds.  This file repeats that exact
// operation sequence on the CPU (std::fma, everything else unfused: -ffp-contract=off), so the
// GPU result can be checked BIT FOR BIT; tests additionally compare it with the faithful CSR
// mat-vec of orc_system.cpp by tolerance (same real-arithmetic value, different rounding).
#include "orc_types.hpp"
#include "tm_oracle.h"
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace orc {

enum { MIRROR_RAW = 0, MIRROR_SCALED = 1, MIRROR_RESID = 2, MIRROR_RELAX = 3 };

static inline Vec2d mirror_row(int mode, bool has_pq, Vec2d m_l, Vec2d m_c, Vec2d m_r, Vec2d c_l, Vec2d c_c, Vec2d c_r, Vec2d p_l, Vec2d p_c,
                               Vec2d p_r, Vec2d xm, Vec2d xp, Vec2d xl, Vec2d xr, Float P, Float Q, Float omega) {
    const Float dxi_x = xp.data[0] - xm.data[0], dxi_y = xp.data[1] - xm.data[1];
    const Float det_x = xr.data[0] - xl.data[0], det_y = xr.data[1] - xl.data[1];
    const Float G11 = std::fma(dxi_x, dxi_x, dxi_y * dxi_y);
    const Float G22 = std::fma(det_x, det_x, det_y * det_y);
    const Float G12 = std::fma(dxi_x, det_x, dxi_y * det_y);
    const Float D = G11 + G22;
    const Float m2D = -2.0 * D;
    const Float mhG12 = -0.5 * G12;
    Float ax = p_c.data[0] + m_c.data[0], ay = p_c.data[1] + m_c.data[1];
    Float bx = c_r.data[0] + c_l.data[0], by = c_r.data[1] + c_l.data[1];
    if (has_pq) {
        const Float hP = 0.5 * P, hQ = 0.5 * Q;
        ax = std::fma(hP, p_c.data[0] - m_c.data[0], ax);
        ay = std::fma(hP, p_c.data[1] - m_c.data[1], ay);
        bx = std::fma(hQ, c_r.data[0] - c_l.data[0], bx);
        by = std::fma(hQ, c_r.data[1] - c_l.data[1], by);
    }
    // cross term from the per-row differences e = w(j+1) - w(j-1), which the device forms once per loaded row
    const Float kx = (p_r.data[0] - p_l.data[0]) - (m_r.data[0] - m_l.data[0]);
    const Float ky = (p_r.data[1] - p_l.data[1]) - (m_r.data[1] - m_l.data[1]);
    Float sx = G22 * ax, sy = G22 * ay;
    sx = std::fma(G11, bx, sx);
    sy = std::fma(G11, by, sy);
    if (mode == MIRROR_RELAX && omega == 1.0) {   // the device's full Jacobi step: x_new = q / (2 D), q = 4 x the off-diagonal part of the row
        const Float qx = std::fma(mhG12, kx, sx), qy = std::fma(mhG12, ky, sy);
        if (D == 0.0) return c_c;                 // degenerate cell: every coefficient vanishes, the node stays
        const Float r = 1.0 / m2D;
        return vinit(-(qx * r), -(qy * r));
    }
    sx = std::fma(m2D, c_c.data[0], sx);
    sy = std::fma(m2D, c_c.data[1], sy);
    sx = std::fma(mhG12, kx, sx);
    sy = std::fma(mhG12, ky, sy);
    if (mode == MIRROR_RAW) return vinit(0.25 * sx, 0.25 * sy);
    const Float rinv = (D == 0.0) ? 0.25 : 1.0 / m2D;
    const Float tx = sx * rinv, ty = sy * rinv;
    if (mode == MIRROR_SCALED) return vinit(tx, ty);
    if (mode == MIRROR_RESID) return vinit(-tx, -ty);
    const Float dx = omega * (-tx), dy = omega * (-ty);   // the displacement of this sweep (kept unfused like the device)
    return vinit(c_c.data[0] + dx, c_c.data[1] + dy);
}

// interior rows of one block: out(i,j) = row(in; coefficients from xk, pq); perimeter of `out` untouched
static void mirror_apply_rows(int mode, Index i_begin, Index i_end, Index nj, const Vec2d* in, const Vec2d* xk, const Vec2d* pq, Vec2d* out, Float omega) {
    for (Index i = i_begin; i < i_end; ++i)
        for (Index j = 1; j + 1 < nj; ++j) {
            const Index p = i * nj + j;
            const Float P = pq ? pq[p].data[0] : 0.0, Q = pq ? pq[p].data[1] : 0.0;
            out[p] = mirror_row(mode, pq != nullptr, in[p - nj - 1], in[p - nj], in[p - nj + 1], in[p - 1], in[p], in[p + 1], in[p + nj - 1],
                                in[p + nj], in[p + nj + 1], xk[p - nj], xk[p + nj], xk[p - 1], xk[p + 1], P, Q, omega);
        }
}

void mirror_apply_block(int mode, Index ni, Index nj, const Vec2d* in, const Vec2d* xk, const Vec2d* pq, Vec2d* out, Float omega) {
    for (Index i = 1; i + 1 < ni; ++i)
        for (Index j = 1; j + 1 < nj; ++j) {
            const Index p = i * nj + j;
            const Float P = pq ? pq[p].data[0] : 0.0, Q = pq ? pq[p].data[1] : 0.0;
            out[p] = mirror_row(mode, pq != nullptr, in[p - nj - 1], in[p - nj], in[p - nj + 1], in[p - 1], in[p], in[p + 1], in[p + nj - 1],
                                in[p + nj], in[p + nj + 1], xk[p - nj], xk[p + nj], xk[p - 1], xk[p + 1], P, Q, omega);
        }
}

}  // namespace orc

extern "C" {

int orc_mirror_apply_block(int mode, uint64_t ni, uint64_t nj, const double* in, const double* xk, const double* pq, double* out, double omega) {
    if (ni < 3 || nj < 3 || mode < 0 || mode > 3) return ORC_E_ARG;
    orc::mirror_apply_block(mode, ni, nj, reinterpret_cast<const orc::Vec2d*>(in), reinterpret_cast<const orc::Vec2d*>(xk),
                            reinterpret_cast<const orc::Vec2d*>(pq), reinterpret_cast<orc::Vec2d*>(out), omega);
    return ORC_OK;
}

// `sweeps` Jacobi elliptic sweeps of one block with fixed boundary, device operation order
// (cpu_baseline leg of bench.py and the bit-exact relax test).  Result left in xy.
double orc_time_relax_sweeps(uint64_t ni, uint64_t nj, double* xy, double* scratch, uint64_t sweeps, double omega) {
    using clk = std::chrono::steady_clock;
    orc::Vec2d* a = reinterpret_cast<orc::Vec2d*>(xy);
    orc::Vec2d* b = reinterpret_cast<orc::Vec2d*>(scratch);
    std::memcpy(b, a, sizeof(orc::Vec2d) * ni * nj);   // perimeter of the ping-pong buffer (outside the timed region)
    const auto t0 = clk::now();
    for (uint64_t s = 0; s < sweeps; ++s) {
        orc::mirror_apply_block(orc::MIRROR_RELAX, ni, nj, a, a, nullptr, b, omega);
        std::swap(a, b);
    }
    const auto t1 = clk::now();
    if (sweeps % 2 == 1) std::memcpy(xy, scratch, sizeof(orc::Vec2d) * ni * nj);
    return std::chrono::duration<double>(t1 - t0).count();
}

// The same sweeps with the rows of every sweep split over `threads` host threads (NOT the reference's behaviour -- it is
// single-threaded; reported beside the 1-thread figure as "what the host could do").  Same arithmetic, same bits.
double orc_time_relax_sweeps_mt(uint64_t ni, uint64_t nj, double* xy, double* scratch, uint64_t sweeps, double omega, uint32_t threads) {
    using clk = std::chrono::steady_clock;
    orc::Vec2d* a = reinterpret_cast<orc::Vec2d*>(xy);
    orc::Vec2d* b = reinterpret_cast<orc::Vec2d*>(scratch);
    std::memcpy(b, a, sizeof(orc::Vec2d) * ni * nj);
    const uint32_t T = threads ? threads : 1;
    const auto t0 = clk::now();
    for (uint64_t s = 0; s < sweeps; ++s) {
        std::vector<std::thread> pool;
        const uint64_t rows = ni - 2;
        for (uint32_t t = 0; t < T; ++t) {
            const uint64_t r0 = 1 + rows * t / T, r1 = 1 + rows * (t + 1) / T;
            pool.emplace_back([=]() { orc::mirror_apply_rows(orc::MIRROR_RELAX, r0, r1, nj, a, a, nullptr, b, omega); });
        }
        for (auto& th : pool) th.join();
        std::swap(a, b);
    }
    const auto t1 = clk::now();
    if (sweeps % 2 == 1) std::memcpy(xy, scratch, sizeof(orc::Vec2d) * ni * nj);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
