// gfx950 (MI355X, CDNA4) kernels of the structured-grid smoother.  HIP only, wave64 only.
//
// All arithmetic is fp64 and HBM-bandwidth bound (about 2 flop/B): no MFMA.  The file is built
// with -ffp-contract=off: the compiler never fuses a*b+c on its own, so TFI (K1) and the
// perimeter rows (K4/K5) evaluate exactly the reference's strict-IEEE Zig expressions and match
// the CPU oracle bit for bit.  The interior rows (K2) use an algebraically factored form with
// EXPLICIT fma() calls (half the fp64 work of the term-by-term sum); the oracle mirrors that
// sequence (oracle/orc_mirror.cpp), so K2 is bit-exact against the mirror and within rounding
// (tolerance in the tests) of the reference's CSR mat-vec.  Reductions differ by tree order.
//
// K2 design (the dominant kernel).  A workgroup is 4 waves side by side; wave w owns the 64
// columns [j0, j0+64) of a 256-column strip and MARCHES down a chunk of rows, keeping a
// 3-row window of the vector (and of the frozen coordinate field) in registers:
//   - every row is read once per wave with one coalesced 16 B/lane load (1 KiB per wave,
//     128 B-line aligned because strips start at multiples of 64 columns);
//   - the j-1 / j+1 neighbours come from the adjacent lanes by DPP wave shifts
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1), the two strip-edge columns from one extra
//     2-lane load whose cache line is the neighbouring wave's own line (L1/L2 hit);
//   - no LDS tile, no barrier in the row loop, so the four waves stay independent and the
//     loads of U rows are in flight together;
//   - Jacobi scaling, the relaxation update and the partial dot products are fused into the
//     same pass, so one sweep moves the compulsory 32 B/node (field mode, Laplace);
//   - results are written with non-temporal stores: the output is not re-read before the next
//     sweep, and keeping it out of L2 is worth ~25 % on a streaming copy of this footprint.
// Workgroup ids are remapped so that every XCD (own L2) walks a contiguous range of
// (row-chunk, strip) tiles: the halo lines shared by neighbouring tiles hit in that L2.
#include "tm_kernels.h"
#include "tm_devutil.hpp"
#include "tm_refmath.h"
#include <cstdlib>
#include <type_traits>

namespace tmh {

// ------------------------------------------------------------------------------------------
// lane exchange: value of the previous / next lane of the wave; lane 0 / 63 keep `edge`
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_prev(double edge, double v) {
    int2 o = __builtin_bit_cast(int2, edge), s = __builtin_bit_cast(int2, v), r;
    r.x = __builtin_amdgcn_update_dpp(o.x, s.x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    r.y = __builtin_amdgcn_update_dpp(o.y, s.y, 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double lane_next(double edge, double v) {
    int2 o = __builtin_bit_cast(int2, edge), s = __builtin_bit_cast(int2, v), r;
    r.x = __builtin_amdgcn_update_dpp(o.x, s.x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    r.y = __builtin_amdgcn_update_dpp(o.y, s.y, 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double2 lane_prev(double2 edge, double2 v) { return make_double2(lane_prev(edge.x, v.x), lane_prev(edge.y, v.y)); }
__device__ __forceinline__ double2 lane_next(double2 edge, double2 v) { return make_double2(lane_next(edge.x, v.x), lane_next(edge.y, v.y)); }

// zero-fill variants (bound_ctrl): the wave-edge lane receives 0 and no `old` register has to be prepared
__device__ __forceinline__ double2 lane_prev0(double2 v) {
    int4 s = __builtin_bit_cast(int4, v), r;
    r.x = __builtin_amdgcn_update_dpp(0, s.x, 0x138, 0xf, 0xf, true);
    r.y = __builtin_amdgcn_update_dpp(0, s.y, 0x138, 0xf, 0xf, true);
    r.z = __builtin_amdgcn_update_dpp(0, s.z, 0x138, 0xf, 0xf, true);
    r.w = __builtin_amdgcn_update_dpp(0, s.w, 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double2, r);
}
__device__ __forceinline__ double2 lane_next0(double2 v) {
    int4 s = __builtin_bit_cast(int4, v), r;
    r.x = __builtin_amdgcn_update_dpp(0, s.x, 0x130, 0xf, 0xf, true);
    r.y = __builtin_amdgcn_update_dpp(0, s.y, 0x130, 0xf, 0xf, true);
    r.z = __builtin_amdgcn_update_dpp(0, s.z, 0x130, 0xf, 0xf, true);
    r.w = __builtin_amdgcn_update_dpp(0, s.w, 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double2, r);
}

// ------------------------------------------------------------------------------------------
// 9-point Winslow/Poisson coefficients -- reference smooth.zig:171-216 (StencilData.init),
// same expression order.  Slots follow StencilData.index.
// ------------------------------------------------------------------------------------------
enum { S_I_J = 0, S_IP1_J, S_IM1_J, S_I_JP1, S_I_JM1, S_IP1_JP1, S_IP1_JM1, S_IM1_JP1, S_IM1_JM1 };

template <bool HAS_PQ>
__device__ __forceinline__ void stencil_coefs(double2 im1_j, double2 ip1_j, double2 i_jm1, double2 i_jp1, double P, double Q,
                                              double (&c)[9]) {
    const double x_xi = 0.5 * (ip1_j.x - im1_j.x);
    const double x_eta = 0.5 * (i_jp1.x - i_jm1.x);
    const double y_xi = 0.5 * (ip1_j.y - im1_j.y);
    const double y_eta = 0.5 * (i_jp1.y - i_jm1.y);
    const double g22 = x_eta * x_eta + y_eta * y_eta;
    const double g12 = x_xi * x_eta + y_xi * y_eta;
    const double g11 = x_xi * x_xi + y_xi * y_xi;
    c[S_I_J] = -2.0 * g22 - 2.0 * g11;
    if (HAS_PQ) {
        c[S_IP1_J] = g22 * (1 + 0.5 * P);
        c[S_IM1_J] = g22 * (1 - 0.5 * P);
        c[S_I_JP1] = g11 * (1 + 0.5 * Q);
        c[S_I_JM1] = g11 * (1 - 0.5 * Q);
    } else {   // P = Q = 0: g * (1 +- 0) == g exactly
        c[S_IP1_J] = g22;
        c[S_IM1_J] = g22;
        c[S_I_JP1] = g11;
        c[S_I_JM1] = g11;
    }
    c[S_IP1_JP1] = -0.5 * g12;
    c[S_IP1_JM1] = 0.5 * g12;
    c[S_IM1_JP1] = 0.5 * g12;
    c[S_IM1_JM1] = -0.5 * g12;
}

// out value for one row given sum=(A in)_row, the row's rhs and diagonal (both components)
template <int MODE>
__device__ __forceinline__ double row_out(double sum, double rhs, double diag, double in_self, double omega) {
    if (MODE == MODE_RAW) return sum;
    const double dinv = (diag == 0.0) ? 1.0 : 1.0 / diag;   // BiCGStab.zig:169-173
    if (MODE == MODE_SCALED) return sum * dinv;
    const double res = rhs * dinv - sum * dinv;
    if (MODE == MODE_RESID) return res;
    return in_self + omega * res;
}

template <int DOT>
__device__ __forceinline__ void accumulate(double (&acc)[MAX_PARTIALS], double2 in_self, double2 out, double2 aux) {
    if (DOT == DOT_AUX) {
        acc[0] += aux.x * out.x;
        acc[1] += aux.y * out.y;
    } else if (DOT == DOT_AUX2) {
        acc[0] += aux.x * out.x;
        acc[1] += aux.y * out.y;
        acc[2] += out.x * out.x;
        acc[3] += out.y * out.y;
    } else if (DOT == DOT_IN) {
        acc[0] += in_self.x * out.x;
        acc[1] += in_self.y * out.y;
        acc[2] += out.x * out.x;
        acc[3] += out.y * out.y;
    } else if (DOT == DOT_IN_SS) {
        acc[0] += in_self.x * out.x;
        acc[1] += in_self.y * out.y;
        acc[2] += out.x * out.x;
        acc[3] += out.y * out.y;
        acc[4] += in_self.x * in_self.x;
        acc[5] += in_self.y * in_self.y;
    } else if (DOT == DOT_B2) {
        acc[0] += in_self.x * out.x;
        acc[1] += in_self.y * out.y;
        acc[2] += out.x * out.x;
        acc[3] += out.y * out.y;
        acc[4] += aux.x * in_self.x;
        acc[5] += aux.y * in_self.y;
        acc[6] += aux.x * out.x;
        acc[7] += aux.y * out.y;
    } else if (DOT == DOT_OUT2) {
        acc[0] += out.x * out.x;
        acc[1] += out.y * out.y;
    } else if (DOT == DOT_DELTA) {   // interior rows pass the displacement in `aux`
        acc[0] = fma(aux.x, aux.x, acc[0]);
        acc[1] = fma(aux.y, aux.y, acc[1]);
    }
}

// ------------------------------------------------------------------------------------------
// Interior row in FACTORED form.  With the unscaled central differences
//     d_xi = X(i+1,j) - X(i-1,j),  d_eta = X(i,j+1) - X(i,j-1)          (= 2 x_xi, 2 x_eta)
//     G11 = d_xi.d_xi = 4 g11,  G22 = d_eta.d_eta = 4 g22,  G12 = d_xi.d_eta = 4 g12
// the reference's row  sum = sum_k c_k w_k  (coefficients of smooth.zig:171-216) is
//   4 sum = G22 [ (w_p + w_m) + P/2 (w_p - w_m) ] + G11 [ (w_r + w_l) + Q/2 (w_r - w_l) ]
//           - 2 (G11 + G22) w_c - G12/2 [ (w_pp + w_mm) - (w_pm + w_mp) ],      4 a_ii = -2 (G11 + G22)
// (m/c/p = rows i-1/i/i+1, l/r = columns j-1/j+1).  About half the fp64 operations of the
// term-by-term CSR sum; same real-arithmetic value, rounding differs at the 1e-16 level, so
// interior rows are compared with the faithful CSR mat-vec by tolerance and BIT-EXACTLY with the
// oracle's mirror of this very sequence (oracle/orc_mirror.cpp): every fma below is explicit,
// everything else is built with -ffp-contract=off.
// ------------------------------------------------------------------------------------------
// rinv = (D == 0) ? 0.25 : 1.0 / m2D, correctly rounded (the oracle's mirror divides with IEEE `/`).
// hipcc expands an f64 division into div_scale x2, rcp, the Newton/Markstein chain (5 fma + 1 mul), div_fmas and div_fixup.
// With numerator 1 and a denominator whose exponent is far from the ends of the range, the scalings are the identity and
// the fix-up passes the value through, so the chain alone -- v_rcp_f64 + 6 fma -- returns the same bits; that covers every
// mesh whose squared spacings lie in [2^-700, 2^700].  Anything else (D == 0, subnormal / huge / non-finite spacings) takes
// the full division; the choice is wave-uniform, so the common case pays two compares and no select.
__device__ __forceinline__ double recip_diag(double D, double m2D) {
    const bool plain = (D >= 0x1p-700) && (D <= 0x1p700);
    if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
        const double nd = -m2D;
        const double r0 = __builtin_amdgcn_rcp(m2D);
        const double e0 = fma(nd, r0, 1.0);
        const double r1 = fma(r0, e0, r0);
        const double e1 = fma(nd, r1, 1.0);
        const double r2 = fma(r1, e1, r1);
        const double e2 = fma(nd, r2, 1.0);   // residual of the quotient 1 * r2
        return fma(e2, r2, r2);
    }
    return (D == 0.0) ? 0.25 : 1.0 / m2D;
}

// A row of the vector enters as (c, e, h): the value and, from the two lane shifts done once when the row is loaded,
// e = w(i,j+1) - w(i,j-1) and h = w(i,j+1) + w(i,j-1).  The cross term is k = e(i+1) - e(i-1).
// dxi = xk(i+1,j) - xk(i-1,j), det = xk(i,j+1) - xk(i,j-1): the frozen field's differences (field mode: p_c - m_c and e_c).
template <int MODE, bool HAS_PQ, bool UNIT_OMEGA = false>
__device__ __forceinline__ double2 winslow_row(double2 m_c, double2 e_m, double2 c_c, double2 e_c, double2 h_c, double2 p_c, double2 e_p, double2 dxi,
                                               double2 det, double P, double Q, double omega, double2& delta) {
    const double G11 = fma(dxi.x, dxi.x, dxi.y * dxi.y);
    const double G22 = fma(det.x, det.x, det.y * det.y);
    const double G12 = fma(dxi.x, det.x, dxi.y * det.y);
    const double D = G11 + G22;
    const double m2D = -2.0 * D;      // 4 a_ii
    const double mhG12 = -0.5 * G12;
    double ax = p_c.x + m_c.x, ay = p_c.y + m_c.y;
    double bx = h_c.x, by = h_c.y;
    if (HAS_PQ) {
        const double hP = 0.5 * P, hQ = 0.5 * Q;
        ax = fma(hP, p_c.x - m_c.x, ax);
        ay = fma(hP, p_c.y - m_c.y, ay);
        bx = fma(hQ, e_c.x, bx);
        by = fma(hQ, e_c.y, by);
    }
    const double kx = e_p.x - e_m.x;
    const double ky = e_p.y - e_m.y;
    double sx = G22 * ax, sy = G22 * ay;
    sx = fma(G11, bx, sx);
    sy = fma(G11, by, sy);
    if (MODE == MODE_RELAX && (UNIT_OMEGA || omega == 1.0)) {
        // A full Jacobi step in its textbook form x_new = (b - sum_{k != i} a_ik x_k) / a_ii = q / (2 D), q = 4 x the off-diagonal part of
        // the row: the diagonal term and the "c + displacement" addition of the general form below cancel analytically, which
        // saves 4 of ~52 fp64 operations per node and sweep -- and the sweep is bound by fp64 issue under the board's power cap
        // (DESIGN.md section 4).  The oracle's mirror makes the same step (oracle/orc_mirror.cpp); omega != 1 keeps the general form.
        const double qx = fma(mhG12, kx, sx), qy = fma(mhG12, ky, sy);
        const bool plain = (D >= 0x1p-700) && (D <= 0x1p700);   // see recip_diag
        double2 xn;
        if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
            const double nd = -m2D;
            const double r0 = __builtin_amdgcn_rcp(m2D);
            const double e0 = fma(nd, r0, 1.0);
            const double r1 = fma(r0, e0, r0);
            const double e1 = fma(nd, r1, 1.0);
            const double r2 = fma(r1, e1, r1);
            const double e2 = fma(nd, r2, 1.0);
            const double r = fma(e2, r2, r2);   // 1 / (-2 D), correctly rounded
            xn = make_double2(-(qx * r), -(qy * r));
        } else {   // a degenerate cell (all coefficients vanish: the row reads 0 = 0, the node stays) or extreme spacings
            const double r = 1.0 / m2D;
            xn = (D == 0.0) ? c_c : make_double2(-(qx * r), -(qy * r));
        }
        delta = make_double2(xn.x - c_c.x, xn.y - c_c.y);
        return xn;
    }
    sx = fma(m2D, c_c.x, sx);
    sy = fma(m2D, c_c.y, sy);
    sx = fma(mhG12, kx, sx);
    sy = fma(mhG12, ky, sy);   // sx, sy = 4 * (A w)_row
    if (MODE == MODE_RAW) return make_double2(0.25 * sx, 0.25 * sy);
    const double rinv = recip_diag(D, m2D);   // 1 / (4 a_ii); a_ii == 0 -> D^-1 := 1 (BiCGStab.zig:169-173)
    const double tx = sx * rinv, ty = sy * rinv;          // (D^-1 A w)_row
    if (MODE == MODE_SCALED) return make_double2(tx, ty);
    if (MODE == MODE_RESID) return make_double2(-tx, -ty);   // b = 0 on interior rows
    // MODE_RELAX: the displacement of this sweep (omega == 1: the product is exact, so it is skipped)
    delta = UNIT_OMEGA ? make_double2(-tx, -ty) : make_double2(omega * (-tx), omega * (-ty));
    return make_double2(c_c.x + delta.x, c_c.y + delta.y);
}

__device__ __forceinline__ double2 sub2(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }

template <int NT>
__device__ __forceinline__ const KrylovScalars* lazy_scalars(const LazyScalars& L);   // defined with the Krylov vector kernels

// ------------------------------------------------------------------------------------------
// K2  winslow_apply: interior rows of one block (replaces smooth.zig:923-992 fill +
//     BiCGStab.zig:424-435 mat-vec), matrix-free, factored row evaluation (winslow_row)
// ------------------------------------------------------------------------------------------
static bool g_rows_forced = false;   // tm_tune_apply chose the chunk length: no per-block rule
static int g_rows_per_chunk = 18;   // tunable (tm_tune_apply); short chunks: see relax2_rows_per_chunk
static int g_unroll = 6;   // rows per load group: 3 or 6
static int g_nt = 1;

constexpr unsigned OOB_VOFFSET = 0x80000000u;   // beyond num_records of every buffer resource below: the hardware drops the lane's store
typedef int v4i32 __attribute__((ext_vector_type(4)));

typedef double d2v __attribute__((ext_vector_type(2)));
// streaming store of a double2 (the vectors are far larger than the caches; worth +25 % on a copy of this footprint)
__device__ __forceinline__ void store_nt(double2* dst, double2 v) {
    d2v o;
    o.x = v.x;
    o.y = v.y;
    __builtin_nontemporal_store(o, reinterpret_cast<d2v*>(dst));
}

// The vector a Krylov update would have stored, formed where it is consumed (the expressions of k_s_update / k_p_update; the
// library is built with -ffp-contract=off, so they give the same bits wherever they are formed)
// VK_R (x = r, y = v, z = t, w = p; va = alpha, vb = omega, vc = beta): the pending x / r update and the p-update in one go
struct VirtualR {
    double2 s, rn, pn;   // s = r - alpha v, r' = s - omega t, p' = r' + beta (p - omega v)
};
__device__ __forceinline__ VirtualR virtual_r(double2 x, double2 y, double2 z, double2 w, double2 va, double2 vb, double2 vc) {
    VirtualR o;
    o.s = make_double2(x.x - va.x * y.x, x.y - va.y * y.y);
    o.rn = make_double2(o.s.x - vb.x * z.x, o.s.y - vb.y * z.y);
    o.pn = make_double2(o.rn.x + vc.x * (w.x - vb.x * y.x), o.rn.y + vc.y * (w.y - vb.y * y.y));
    return o;
}
template <int VK>
__device__ __forceinline__ double2 virtual_vec(double2 x, double2 y, double2 z, double2 va, double2 vb) {
    if (VK == VK_S || VK == VK_S2) return make_double2(x.x - va.x * y.x, x.y - va.y * y.y);   // s = r - alpha v (BiCGStab.zig:325-327)
    if (VK == VK_P) return make_double2(x.x + va.x * (y.x - vb.x * z.x), x.y + va.y * (y.y - vb.y * z.y));   // p = r + beta (p - omega v) (BiCGStab.zig:310-312)
    return x;
}

__device__ __forceinline__ double2 load_nt(const double2* src);   // defined with the vector kernels

// One workgroup's tile of one block; `bid` = the workgroup's index within that block's tiles (also its partial-sum slot).
// VK (virtual input vector, formed as the rows are taken into the window; va, vb component-wise):
//   VK_S: in - va * in2                (s = r - alpha v, never stored)
//   VK_P: in + va * (in2 - vb * in3)   (p = r + beta (p - omega v), stored to a.pout for the owned rows)
//   VK_R: virtual_r(in, in2, in3, in4) (va, vb, vc = alpha, omega, beta): r', p' and the solution update stored for the owned rows
//         as each row ENTERS the window (a row enters once per workgroup; the chunk's two halo rows belong to the neighbours)
// OV (overlapping strips, the Krylov kernels of large meshes): a wave holds the 64 columns 62 s .. 62 s + 63 of strip s and OWNS the 62 in
// the middle (lanes 1 .. 62), so both j-neighbours of every owned column sit in the wave and no halo column is ever loaded -- with seven
// input streams (VK_R) the halo values of a six-row load group were 120 of the kernel's ~290 registers.  3 % of the columns are loaded
// twice (mostly from L2).  Same arithmetic per node, same bits.
template <int MODE, int DOT, bool FIELD, bool HAS_PQ, int U, bool NT, int VK = VK_NONE, bool OV = false>
__device__ __forceinline__ void apply_tile(const ApplyBlock& a, int RI, int nSG, int nRC, int bid, double2 va = make_double2(0.0, 0.0),
                                           double2 vb = make_double2(0.0, 0.0), double2 vc = make_double2(0.0, 0.0)) {
    static_assert(U % 3 == 0, "the 3-row window rotates by renaming: the row group must be a multiple of 3");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware tile order: physical workgroup b runs on XCD b%8; give each XCD a contiguous
    // range of logical tiles (speed only, any placement is correct).
    const int total = nSG * nRC;
    const int q = total >> 3, rem = total & 7, xcd = bid & 7, k = bid >> 3;
    const int logical = (xcd < rem) ? xcd * (q + 1) + k : rem * (q + 1) + (xcd - rem) * q + k;
    const int rc = logical / nSG;
    const int sg = logical - rc * nSG;

    const int ni = a.ni, nj = a.nj;
    const int j0 = (sg * 4 + wave) * (OV ? 62 : 64);   // OV: the wave's first LOADED column (lane 0); its first owned column is j0 + 1
    const int j = j0 + lane;
    const int jc = min(j, nj - 1);
    const bool edge_lane = !OV && ((lane == 0) || (lane == 63));
    const int hcol = (lane == 0) ? max(j0 - 1, 0) : min(j0 + 64, nj - 1);
    const bool valid_col = (j >= 1) && (j <= nj - 2) && (!OV || (lane >= 1 && lane <= 62));
    // wave-uniform: every lane owns an output column (never with OV: its two halo lanes store nothing, so it takes the predicated path --
    // masking them by out-of-range buffer-store offsets instead, as K2x2 does, measured SLOWER here: 4096^2 914 against 876 us per iteration)
    const bool full_wave = !OV && (j0 >= 1) && (j0 + 63 <= nj - 2);
    const int i0 = 1 + rc * RI;
    const int i1 = min(i0 + RI, ni - 1);   // output rows [i0, i1)

    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if ((OV ? (j0 + 1 <= nj - 2) : (j0 < nj)) && i0 < i1) {   // wave-uniform
        auto load_row = [&](const double2* __restrict__ v, int row, double2& c, double2& h) {
            const double2* rp = v + static_cast<size_t>(row) * nj;
            c = rp[jc];
            h = make_double2(0.0, 0.0);   // NOT `h = c`: that copy would force a vmcnt(0) wait right behind every row load
            if (edge_lane) h = rp[hcol];
        };
        // 3-row window of the vector, rotating by index: slot (r % 3) holds row i0-1+r as (value, e = right - left, h = right + left)
        double2 Wc[3], We[3], Wh[3];
        double2 Xc[3], Xdet = make_double2(0.0, 0.0);   // frozen coordinates when they are a different array; Xdet of the centre row
        auto comb = [&](double2 x, double2 y, double2 z) { return virtual_vec<VK>(x, y, z, va, vb); };
        // VK_R: centre value of an entering row -> p'; for an owned row the node's r', p', u are stored and ||r'||^2 accumulated
        auto enter = [&](int row, bool owned_row, double2 x, double2 y, double2 z, double2 w, double2 u0) {
            const VirtualR q = virtual_r(x, y, z, w, va, vb, vc);
            if (owned_row && valid_col) {
                const size_t at = static_cast<size_t>(row) * nj + j;
                double2 un = u0;
                un.x += va.x * w.x;   // BiCGStab.zig:329-331 (x += alpha * p_hat), k_xr_update_vs's expressions
                un.y += va.y * w.y;
                un.x += vb.x * q.s.x;   // BiCGStab.zig:352-354 (x += omega * s_hat)
                un.y += vb.y * q.s.y;
                store_nt(a.uio + at, un);
                store_nt(a.rout + at, q.rn);
                store_nt(a.pout + at, q.pn);
                acc[2] += q.rn.x * q.rn.x;
                acc[3] += q.rn.y * q.rn.y;
            }
            return q.pn;
        };
        auto halo_r = [&](double2 x, double2 y, double2 z, double2 w) { return virtual_r(x, y, z, w, va, vb, vc).pn; };
        // VK_PRO: e' = e + bilinear interpolation of the coarse correction (k_mg_prolong_add).  Per fine row two coarse rows (ci0,
        // ci1); per lane the coarse column cj0 of its own column -- the column cj1 is the right neighbour's cj0 when the fine column
        // is odd (lane shift), its own otherwise.  Edge lanes fetch one more coarse value for their halo column.
        const int pro_cj0 = (VK == VK_PRO) ? (a.mg_cj ? (jc >> 1) : jc) : 0;
        const int pro_hc = (VK == VK_PRO) ? min(a.mg_cj ? ((lane == 0) ? ((j0 - 1) >> 1) : ((j0 + 64) >> 1)) : hcol, a.mg_njc - 1) : 0;
        const bool pro_odd = (VK == VK_PRO) && a.mg_cj && (jc & 1);
        const bool pro_col = (j >= 1) && (j <= nj - 2);
        const bool pro_hcol = (lane == 0) ? (j0 - 1 >= 1) : (j0 + 64 <= nj - 2);
        auto load_coarse = [&](int row, int which, double2& c, double2& h) {   // which = 0: coarse row ci0, 1: ci1
            const int cr = min(a.mg_ci ? ((row + which) >> 1) : row, a.mg_nic - 1);
            const double2* rp = a.in2 + static_cast<size_t>(cr) * a.mg_njc;
            c = rp[max(pro_cj0, 0) < a.mg_njc ? max(pro_cj0, 0) : a.mg_njc - 1];
            h = make_double2(0.0, 0.0);
            if (edge_lane) h = rp[max(pro_hc, 0)];
        };
        // e (centre, halo) of fine row `row` + the interpolated correction
        auto prolonged = [&](int row, double2 ec, double2 eh, double2 c0, double2 h0, double2 c1, double2 h1, double2& out_c, double2& out_h) {
            const bool row_in = (row >= 1) && (row <= ni - 2);
            // centre: a = (ci0, cj0), b = (ci0, cj1), c = (ci1, cj0), d = (ci1, cj1)
            const double2 n0 = lane_next(h0, c0), n1 = lane_next(h1, c1);   // shifts first, with every lane active; then the select
            const double2 b0 = pro_odd ? n0 : c0, b1 = pro_odd ? n1 : c1;
            out_c = ec;
            if (row_in && pro_col) {
                out_c.x += 0.25 * ((c0.x + b0.x) + (c1.x + b1.x));
                out_c.y += 0.25 * ((c0.y + b0.y) + (c1.y + b1.y));
            }
            // halo column (edge lanes): left = j0 - 1 (odd when coarsened: a = its own coarse column, b = lane 0's), right = j0 + 64 (even)
            out_h = eh;
            if (edge_lane && row_in && pro_hcol) {
                const bool left_odd = (lane == 0) && a.mg_cj;
                const double2 a0 = h0, a1 = h1, q0 = left_odd ? c0 : h0, q1 = left_odd ? c1 : h1;
                out_h.x += 0.25 * ((a0.x + q0.x) + (a1.x + q1.x));
                out_h.y += 0.25 * ((a0.y + q0.y) + (a1.y + q1.y));
            }
        };
        {
            double2 h0, h1;
            load_row(a.in, i0 - 1, Wc[0], h0);
            load_row(a.in, i0, Wc[1], h1);
            if (VK == VK_PRO) {
                double2 a0, b0, a1, b1, c0, d0, c1, d1;
                load_coarse(i0 - 1, 0, a0, b0);
                load_coarse(i0 - 1, 1, a1, b1);
                load_coarse(i0, 0, c0, d0);
                load_coarse(i0, 1, c1, d1);
                prolonged(i0 - 1, Wc[0], h0, a0, b0, a1, b1, Wc[0], h0);
                prolonged(i0, Wc[1], h1, c0, d0, c1, d1, Wc[1], h1);
            } else if (VK == VK_R) {
                double2 q0, g0, q1, g1, z0, y0, z1, y1, w0, k0, w1, k1;
                load_row(a.in2, i0 - 1, q0, g0);
                load_row(a.in2, i0, q1, g1);
                load_row(a.in3, i0 - 1, z0, y0);
                load_row(a.in3, i0, z1, y1);
                load_row(a.in4, i0 - 1, w0, k0);
                load_row(a.in4, i0, w1, k1);
                const double2 u1 = a.uio[static_cast<size_t>(i0) * nj + jc];
                h0 = halo_r(h0, g0, y0, k0);
                h1 = halo_r(h1, g1, y1, k1);
                Wc[0] = enter(i0 - 1, false, Wc[0], q0, z0, w0, u1);
                Wc[1] = enter(i0, true, Wc[1], q1, z1, w1, u1);
            } else if (VK != VK_NONE) {
                double2 q0, g0, q1, g1, z0 = make_double2(0.0, 0.0), y0 = z0, z1 = z0, y1 = z0;
                load_row(a.in2, i0 - 1, q0, g0);
                load_row(a.in2, i0, q1, g1);
                if (VK == VK_P) {
                    load_row(a.in3, i0 - 1, z0, y0);
                    load_row(a.in3, i0, z1, y1);
                }
                Wc[0] = comb(Wc[0], q0, z0);
                h0 = comb(h0, g0, y0);
                Wc[1] = comb(Wc[1], q1, z1);
                h1 = comb(h1, g1, y1);
            }
            const double2 l0 = lane_prev(h0, Wc[0]), r0 = lane_next(h0, Wc[0]), l1 = lane_prev(h1, Wc[1]), r1 = lane_next(h1, Wc[1]);
            We[0] = sub2(r0, l0); Wh[0] = add2(r0, l0);
            We[1] = sub2(r1, l1); Wh[1] = add2(r1, l1);
            Wc[2] = We[2] = Wh[2] = make_double2(0.0, 0.0);
            Xc[0] = Xc[1] = Xc[2] = make_double2(0.0, 0.0);
            if (!FIELD) {
                double2 t_h;
                load_row(a.xk, i0 - 1, Xc[0], t_h);
                load_row(a.xk, i0, Xc[1], t_h);
                Xdet = sub2(lane_next(t_h, Xc[1]), lane_prev(t_h, Xc[1]));
            }
        }

        struct Group {
            double2 pc[U], ph[U], xpc[U], xph[U], pqv[U], auxv[U], qc[VK != VK_NONE ? U : 1], qh[VK != VK_NONE ? U : 1];
            double2 zc[(VK == VK_P || VK == VK_R || VK == VK_PRO) ? U : 1], zh[(VK == VK_P || VK == VK_R || VK == VK_PRO) ? U : 1], wc[VK == VK_R ? U : 1], wh[VK == VK_R ? U : 1], uc[VK == VK_R ? U : 1];
        };
        auto load_group = [&](int ib, Group& g) {   // rows ib+1 .. ib+U of the vector (and of xk), pq/aux of rows ib .. ib+U-1
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int prow = min(ib + u + 1, ni - 1);
                if (MODE == MODE_DIAG_NOLOAD || MODE == MODE_DIAG_MATH) {
                    g.pc[u] = Wc[1];
                    g.ph[u] = Wh[1];
                } else
                    load_row(a.in, prow, g.pc[u], g.ph[u]);
                if (VK == VK_PRO) {   // the two coarse rows of this fine row (qc/qh: ci0, zc/zh: ci1)
                    load_coarse(prow, 0, g.qc[u], g.qh[u]);
                    load_coarse(prow, 1, g.zc[u], g.zh[u]);
                } else if (VK != VK_NONE) {
                    load_row(a.in2, prow, g.qc[u], g.qh[u]);
                }
                if (VK == VK_P || VK == VK_R) load_row(a.in3, prow, g.zc[u], g.zh[u]);
                if (VK == VK_R) {
                    load_row(a.in4, prow, g.wc[u], g.wh[u]);
                    g.uc[u] = a.uio[static_cast<size_t>(prow) * nj + jc];
                }
                if (!FIELD) load_row(a.xk, prow, g.xpc[u], g.xph[u]);
                const size_t cur = static_cast<size_t>(min(ib + u, ni - 2)) * nj + jc;
                if (HAS_PQ) g.pqv[u] = a.pq[cur];
                if (DOT == DOT_AUX || DOT == DOT_AUX2 || DOT == DOT_B2 || MODE == MODE_MG_RESID || MODE == MODE_MG_SMOOTH) g.auxv[u] = a.aux[cur];
            }
        };
        // PRED = false: full wave and full row group -> no exec masking around the arithmetic and the store
        auto compute_group = [&](int ib, const Group& g, auto pred_tag) {
            constexpr bool PRED = decltype(pred_tag)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int M = u % 3, C = (u + 1) % 3, P3 = (u + 2) % 3;   // window slots of rows i-1, i, i+1
                const int row = ib + u;
                double2 nc, nh;
                if (VK == VK_PRO) {
                    prolonged(row + 1, g.pc[u], g.ph[u], g.qc[u], g.qh[u], g.zc[VK == VK_PRO ? u : 0], g.zh[VK == VK_PRO ? u : 0], nc, nh);
                } else if (VK == VK_R) {
                    nc = enter(row + 1, row + 1 < i1, g.pc[u], g.qc[u], g.zc[VK == VK_R ? u : 0], g.wc[VK == VK_R ? u : 0], g.uc[VK == VK_R ? u : 0]);
                    nh = halo_r(g.ph[u], g.qh[u], g.zh[VK == VK_R ? u : 0], g.wh[VK == VK_R ? u : 0]);
                } else {
                    nc = VK != VK_NONE ? comb(g.pc[u], g.qc[u], g.zc[VK == VK_P ? u : 0]) : g.pc[u];
                    nh = VK != VK_NONE ? comb(g.ph[u], g.qh[u], g.zh[VK == VK_P ? u : 0]) : g.ph[u];
                }
                Wc[P3] = nc;
                {
                    const double2 l = lane_prev(nh, nc), r = lane_next(nh, nc);
                    We[P3] = sub2(r, l);
                    Wh[P3] = add2(r, l);
                }
                if (!FIELD) Xc[P3] = g.xpc[u];

                const double P = HAS_PQ ? g.pqv[u].x : 0.0, Q = HAS_PQ ? g.pqv[u].y : 0.0;
                double2 o, delta = make_double2(0.0, 0.0);
                if (MODE == MODE_DIAG_COPY) {
                    o = Wc[C];
                } else if (MODE == MODE_DIAG_SUM9) {
                    o.x = Wh[M].x + Wc[M].x + Wh[C].x + Wc[C].x + Wh[P3].x + Wc[P3].x;
                    o.y = Wh[M].y + Wc[M].y + Wh[C].y + Wc[C].y + Wh[P3].y + Wc[P3].y;
                } else if (MODE == MODE_DIAG_NOSTORE || MODE == MODE_DIAG_NOLOAD || MODE == MODE_DIAG_MATH) {
                    o = winslow_row<MODE_RELAX, HAS_PQ>(Wc[M], We[M], Wc[C], We[C], Wh[C], Wc[P3], We[P3], sub2(Wc[P3], Wc[M]), We[C], P, Q, a.omega, delta);
                } else if (MODE == MODE_MG_FIRST2) {   // e1 = omega f, e2 = e1 + omega (f - D^-1 A e1) = omega (2 f - omega D^-1 A f), one pass over f
                    const double2 t = winslow_row<MODE_SCALED, HAS_PQ>(Wc[M], We[M], Wc[C], We[C], Wh[C], Wc[P3], We[P3], sub2(Xc[P3], Xc[M]), Xdet, P, Q, 0.0,
                                                                       delta);
                    o = make_double2(a.omega * fma(-a.omega, t.x, 2.0 * Wc[C].x), a.omega * fma(-a.omega, t.y, 2.0 * Wc[C].y));
                } else if (MODE == MODE_MG_RESID || MODE == MODE_MG_SMOOTH) {   // error equation of a multigrid level: frozen field in xk, rhs in aux
                    const double2 t = winslow_row<MODE_SCALED, HAS_PQ>(Wc[M], We[M], Wc[C], We[C], Wh[C], Wc[P3], We[P3], sub2(Xc[P3], Xc[M]), Xdet, P, Q, 0.0,
                                                                       delta);
                    const double2 res = sub2(g.auxv[u], t);
                    if (MODE == MODE_MG_RESID) {   // UNscaled residual a_ii (f - D^-1 A e): what the restriction averages (see k_mg_restrict)
                        const double2 dxi = sub2(Xc[P3], Xc[M]);
                        const double aii = -0.5 * (fma(dxi.x, dxi.x, dxi.y * dxi.y) + fma(Xdet.x, Xdet.x, Xdet.y * Xdet.y));
                        o = make_double2(aii * res.x, aii * res.y);
                    } else {
                        o = make_double2(fma(a.omega, res.x, Wc[C].x), fma(a.omega, res.y, Wc[C].y));
                    }
                } else if (FIELD) {
                    o = winslow_row<MODE, HAS_PQ>(Wc[M], We[M], Wc[C], We[C], Wh[C], Wc[P3], We[P3], sub2(Wc[P3], Wc[M]), We[C], P, Q, a.omega, delta);
                } else {
                    o = winslow_row<MODE, HAS_PQ>(Wc[M], We[M], Wc[C], We[C], Wh[C], Wc[P3], We[P3], sub2(Xc[P3], Xc[M]), Xdet, P, Q, a.omega, delta);
                }
                if (MODE == MODE_DIAG_NOSTORE || MODE == MODE_DIAG_MATH) {
                    acc[0] += o.x;
                    acc[1] += o.y;
                } else if (!PRED || (row < i1 && valid_col)) {
                    double2* dst = a.out + static_cast<size_t>(row) * nj + j;
                    if (NT) {
                        d2v ov;
                        ov.x = o.x;
                        ov.y = o.y;
                        __builtin_nontemporal_store(ov, reinterpret_cast<d2v*>(dst));
                    } else {
                        *dst = o;
                    }
                    if (VK == VK_P) store_nt(a.pout + static_cast<size_t>(row) * nj + j, Wc[C]);
                    accumulate<DOT>(acc, Wc[C], o, (DOT == DOT_AUX || DOT == DOT_AUX2 || DOT == DOT_B2) ? g.auxv[u] : ((DOT == DOT_DELTA) ? delta : o));
                }
                if (!FIELD) Xdet = sub2(lane_next(g.xph[u], g.xpc[u]), lane_prev(g.xph[u], g.xpc[u]));   // of the next centre row (= row i+1)
            }
        };

        int ib = i0;
        if (full_wave) {
            for (; ib + U <= i1; ib += U) {
                Group g;
                load_group(ib, g);
                compute_group(ib, g, std::false_type{});
            }
        }
        for (; ib < i1; ib += U) {   // partial waves (block edges) and the row tail
            Group g;
            load_group(ib, g);
            compute_group(ib, g, std::true_type{});
        }
    }
    if ((MODE == MODE_DIAG_NOSTORE || MODE == MODE_DIAG_MATH) && acc[0] + acc[1] == 123.456) a.out[0] = make_double2(acc[0], acc[1]);   // keep the arithmetic live
    if (DOT != DOT_NONE) block_partials<256, (VK == VK_R ? 4 : dot_columns(DOT))>(acc, a.partials + static_cast<size_t>(bid) * MAX_PARTIALS);
}

template <int MODE, int DOT, bool FIELD, bool HAS_PQ, int U, bool NT>
__global__ __launch_bounds__(256) void k_apply(ApplyBlock a, int RI, int nSG, int nRC) {
    apply_tile<MODE, DOT, FIELD, HAS_PQ, U, NT>(a, RI, nSG, nRC, blockIdx.x);
}

// The same for up to APPLY_BATCH_MAX blocks in ONE launch: a multi-block mesh on one GPU (the O4H examples: 8 small blocks) is
// launch-bound, not bandwidth-bound.  start[k] = first workgroup of block k.
template <int MODE, int DOT, bool FIELD, bool HAS_PQ, int U, bool NT>
__global__ __launch_bounds__(256) void k_apply_batch(ApplyBatch B) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < APPLY_BATCH_MAX; ++q)
        if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
    apply_tile<MODE, DOT, FIELD, HAS_PQ, U, NT>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], static_cast<int>(blockIdx.x) - B.start[k]);
}


// Multigrid: the first post-smoothing sweep of a level with the prolongation folded in (VK_PRO, tm_kernels.h)
template <bool HAS_PQ>
__global__ __launch_bounds__(256) void k_apply_prolong_smooth(ApplyBlock a, int RI, int nSG, int nRC) {
    apply_tile<MODE_MG_SMOOTH, DOT_NONE, false, HAS_PQ, 3, true, VK_PRO>(a, RI, nSG, nRC, blockIdx.x);
}
static inline int rows_per_chunk(int ni, int nj, int rows);
hipError_t launch_mg_prolong_smooth(const ApplyBlock& a0, const MgPair& g, hipStream_t st) {
    if (a0.ni < 3 || a0.nj < 3) return hipSuccess;
    ApplyBlock a = a0;
    a.mg_nic = g.nic;
    a.mg_njc = g.njc;
    a.mg_ci = g.ci;
    a.mg_cj = g.cj;
    const int RI = rows_per_chunk(a.ni, a.nj, a.rows);
    const int nSG = (a.nj + 255) / 256;
    const int nRC = (a.ni - 2 + RI - 1) / RI;
    const dim3 grid(nSG * nRC), block(256);
    if (a.pq) hipLaunchKernelGGL(k_apply_prolong_smooth<true>, grid, block, 0, st, a, RI, nSG, nRC);
    else hipLaunchKernelGGL(k_apply_prolong_smooth<false>, grid, block, 0, st, a, RI, nSG, nRC);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2x2: two fused Jacobi sweeps per pass (temporal blocking in registers).
//   a  = X^k      rows i, i+1, i+2  (3-row window, one coalesced load per row)
//   s1 = X^(k+1)  rows i-1, i, i+1  (computed from `a`; perimeter values come from `mid`)
//   s2 = X^(k+2)  row i             (computed from s1, stored)
// A wave loads the 64 columns c0-2 .. c0+61 and owns the 60 output columns c0 .. c0+59
// (c0 = 60*strip: every stored segment is a whole number of 64 B sectors); s1 is valid on lanes
// 1..62, s2 on lanes 2..61, so the j-halo comes with the row load itself and no edge-lane loads
// are needed.  Row chunks overlap by two rows on each side.  HBM traffic per PAIR of sweeps is
// one read (x 64/60 x (RI+4)/RI, the overlap mostly hits L2) and one write of the field.
// ------------------------------------------------------------------------------------------
struct Relax2Tile {
    int c0, c, i0, i1;   // first owned column, this lane's column, owned rows [i0, i1)
    bool out_lane;       // lanes 2..61 on interior columns: store s2 (and the ring of s1)
};

#ifndef TM_R2_SAUX
#define TM_R2_SAUX 2   // cache policy of K2x2's result stores: 2 = nt (streaming); experiment builds override it (tools/dev/ab_saux.sh)
#endif
// A row in window form: value, e = right - left, h = right + left (zero-fill shifts: lanes 0 / 63 hold garbage there)
struct Row3 {
    double2 c, e, h;
};
__device__ __forceinline__ Row3 make_row(double2 c) {
    const double2 l = lane_prev0(c), r = lane_next0(c);
    Row3 w;
    w.c = c;
    w.e = sub2(r, l);
    w.h = add2(r, l);
    return w;
}
// one Jacobi sweep at the centre row of (m, c, p), field mode
template <bool W1>
__device__ __forceinline__ double2 relax_row(const Row3& m, const Row3& c, const Row3& p, double omega, double2& delta) {
    return winslow_row<MODE_RELAX, false, W1>(m.c, m.e, c.c, c.e, c.h, p.c, p.e, sub2(p.c, m.c), c.e, 0.0, 0.0, omega, delta);
}

// One wave, one 60-column strip, rows [i0, i1): the strips along the block edge (and short last chunks).  Perimeter values of
// X^(k+1) come from `mid`, the first-interior ring of X^(k+1) goes to it; every row / column index is clamped.  Written like the
// inside path below -- straight-line steps, loads one group ahead, every predicate folded into a select or into the offset of
// a buffer store -- because a workgroup on the edge that waits out the full memory latency at every step would still be running
// long after the rest of the launch has drained.
template <int DOT, int U, int NT, bool W1>
__device__ __forceinline__ void relax2_strip_edge(const Relax2Block& a, const Relax2Tile& t, double (&acc)[MAX_PARTIALS]) {
    const int ni = a.ni, nj = a.nj;
    const int cc = min(max(t.c, 0), nj - 1);
    const double2* in_col = a.in + cc;
    const double2* mid_col = a.mid + cc;
    const double2 zero = make_double2(0.0, 0.0);
    auto load_in = [&](int row) { return in_col[static_cast<size_t>(min(max(row, 0), ni - 1)) * nj]; };
    auto load_mid = [&](int row) { return mid_col[static_cast<size_t>(min(max(row, 0), ni - 1)) * nj]; };
    const bool perim_col = (t.c == 0) || (t.c == nj - 1);
    const bool ring_col = (t.c == 1) || (t.c == nj - 2);
    const int nrows = t.i1 - t.i0;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + static_cast<size_t>(t.i0) * nj, 0, nrows * nj * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t mid_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.mid + static_cast<size_t>(t.i0) * nj, 0, nrows * nj * 16, 0x00020000);
    const unsigned lane_off = static_cast<unsigned>(cc) * 16u;

    Row3 A[3], S[3];
    double2 pa[U], na[U], pm;
    // step s handles row i = i0 - 2 + s: takes a[i+2], forms s1[i+1], then s2[i]; a[i], a[i+1] are preloaded.
    // pa / na: rows a[i+2] of this group's / the next group's steps; pm: X^(k+1) perimeter value of the next step's row
    {
        const double2 a0 = load_in(t.i0 - 2), a1 = load_in(t.i0 - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) pa[u] = load_in(t.i0 + u);
        pm = load_mid(t.i0 - 1);
        A[0] = make_row(a0);
        A[1] = make_row(a1);
        A[2].c = A[2].e = A[2].h = zero;
        S[0] = S[1] = S[2] = A[2];
    }
    const int nsteps = nrows + 2;
    auto group = [&](const int tb) {
#pragma unroll
        for (int u = 0; u < U; ++u) na[u] = load_in(t.i0 + tb + U + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int A0 = u % 3, A1 = (u + 1) % 3, A2 = (u + 2) % 3;   // slots of a[i], a[i+1], a[i+2] and of s1[i-1], s1[i], s1[i+1]
            const int i = t.i0 - 2 + tb + u;
            const int q = i + 1;   // the s1 row formed in this step
            A[A2] = make_row(pa[u]);
            const double2 pm_now = pm;
            pm = load_mid(q + 1);   // one step ahead
            // ---- s1[q]: first sweep at row q; on the perimeter the perimeter-row kernel's value
            double2 d1 = zero;
            double2 s1 = relax_row<W1>(A[A0], A[A1], A[A2], a.omega, d1);
            const bool perim_row = (q <= 0) || (q >= ni - 1);
            if (perim_row || perim_col) s1 = pm_now;   // a select: both values are in registers
            // this wave's chunk owns row q of s1: leave the first-interior ring of X^(k+1) (all other lanes: offset out of range)
            const bool ring = t.out_lane && !perim_row && (q >= t.i0) && (q < t.i1) && ((q == 1) || (q == ni - 2) || ring_col);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, s1), mid_rsrc,
                                                   static_cast<int>(ring ? lane_off + static_cast<unsigned>((q - t.i0) * nj * 16) : OOB_VOFFSET), 0, 0);
            S[A2] = make_row(s1);
            // ---- s2[i]: second sweep at row i (not in the two warm-up steps, not past a short chunk)
            double2 d2 = zero;
            const double2 o = relax_row<W1>(S[A0], S[A1], S[A2], a.omega, d2);
            const bool live = t.out_lane && (i >= t.i0) && (i < t.i1);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, o), out_rsrc,
                                                   static_cast<int>(live ? lane_off + static_cast<unsigned>((i - t.i0) * nj * 16) : OOB_VOFFSET), 0, (NT & 1) ? TM_R2_SAUX : 0);
            if (DOT == DOT_DELTA) {
                if (!live) d2 = zero;
                accumulate<DOT_DELTA>(acc, S[A1].c, o, d2);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pa[u] = na[u];
    };
    group(0);   // peeled: see relax2_strip_inside
    for (int tb = U; tb < nsteps; tb += U) group(tb);
}

// The same strip strictly inside the block (no perimeter row / column, no ring, RI = i1 - i0 a multiple of U): the steady state.
// The loop body is ONE basic block -- gfx9 counts loads and stores in a single in-order counter (vmcnt) and the compiler merges
// the counter state wherever control flow joins, so any branch around a load or store makes every wait for a prefetched row
// also wait for the stores issued since.  Hence: lane predication of the store by a buffer store whose masked lanes carry an
// out-of-range offset (dropped by the hardware), no row predicates, the next group's rows requested before this group's
// arithmetic, and the first group peeled so that both loop entries see the same queue of outstanding operations.
template <int DOT, int U, int NT, bool W1>
__device__ __forceinline__ void relax2_strip_inside(const Relax2Block& a, const Relax2Tile& t, double (&acc)[MAX_PARTIALS]) {
    const int nj = a.nj;
    const double2* in_col = a.in + t.c;
    const int last_row = t.i1 + 1;   // last row of `in` this strip needs; prefetches past it re-read it
    // NT & 2: streaming loads -- the interior pass of a multi-rank sweep pair on a small block, so that it stops evicting what the
    // chain's kernels re-read (costs the pass its own L2 hits on the rows neighbouring chunks share: see DESIGN.md section 6)
    auto load_in = [&](int row) {
        const double2* src = in_col + static_cast<size_t>(min(row, last_row)) * nj;
        return (NT & 2) ? load_nt(src) : *src;
    };
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out + static_cast<size_t>(t.i0) * nj, 0, (t.i1 - t.i0) * nj * 16, 0x00020000);
    const unsigned voff = t.out_lane ? static_cast<unsigned>(t.c) * 16u : OOB_VOFFSET;

    Row3 A[3], S[3];
    double2 pc[U], pn[U];   // rows a[i+2] of this group's steps / of the next group's (requested one group ahead)
    {   // warm-up: s1[i0-1], s1[i0] from a[i0-2 .. i0+1]
        const double2 am2 = load_in(t.i0 - 2), am1 = load_in(t.i0 - 1), a0 = load_in(t.i0), a1 = load_in(t.i0 + 1);
#pragma unroll
        for (int u = 0; u < U; ++u) pc[u] = load_in(t.i0 + 2 + u);
        const Row3 rm2 = make_row(am2), rm1 = make_row(am1);
        A[0] = make_row(a0);
        A[1] = make_row(a1);
        double2 d = make_double2(0.0, 0.0);
        S[0] = make_row(relax_row<W1>(rm2, rm1, A[0], a.omega, d));
        S[1] = make_row(relax_row<W1>(rm1, A[0], A[1], a.omega, d));
        A[2].c = A[2].e = A[2].h = make_double2(0.0, 0.0);
        S[2] = A[2];
    }
    auto group = [&](const int tb) {   // rows i0+tb .. i0+tb+U-1
#pragma unroll
        for (int u = 0; u < U; ++u) {
#if defined(TM_R2_DIAG) && (TM_R2_DIAG & 1)
            pn[u] = make_double2(pc[u].x * 1.0000001, pc[u].y);   // diagnostic build: no loads in the steady state
#else
            pn[u] = load_in(t.i0 + tb + U + 2 + u);
#endif
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int A0 = u % 3, A1 = (u + 1) % 3, A2 = (u + 2) % 3;   // slots of a[i], a[i+1], a[i+2] and of s1[i-1], s1[i], s1[i+1]
            A[A2] = make_row(pc[u]);
            double2 d1 = make_double2(0.0, 0.0), d2 = make_double2(0.0, 0.0);
            S[A2] = make_row(relax_row<W1>(A[A0], A[A1], A[A2], a.omega, d1));
            const double2 o = relax_row<W1>(S[A0], S[A1], S[A2], a.omega, d2);
            // the row offset travels in the VGPR offset: with an SGPR soffset the 128-bit store data was observed to be picked
            // up late (lanes 12..15 of each 16 saw a later VALU result) -- measured on gfx950, tools/dev/dbg_fused.py
#if defined(TM_R2_DIAG) && (TM_R2_DIAG & 2)
            acc[2] += o.x + o.y;   // diagnostic build: no stores
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, o), out_rsrc, static_cast<int>(voff + static_cast<unsigned>((tb + u) * nj * 16)), 0,
                                                   (NT & 1) ? TM_R2_SAUX : 0);
#endif
            if (DOT == DOT_DELTA) {
                if (!t.out_lane) d2 = make_double2(0.0, 0.0);   // masked lanes hold garbage (possibly non-finite): select, never multiply
                accumulate<DOT_DELTA>(acc, S[A1].c, o, d2);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pc[u] = pn[u];
    };
    group(0);
    for (int tb = U; tb < t.i1 - t.i0; tb += U) group(tb);
}

// Row chunks of a K2x2 block: [R2_HEAD rows][RI rows] x nf [leftover][R2_HEAD rows].  The first and the last chunk are short
// because the workgroups that own them are the ones a multi-rank pass has to run AFTER the halo exchange, one serial chain of
// steps each (23 us with 18 rows, measured): the shorter that chain, the sooner the next pair's interior pass starts.
constexpr int R2_HEAD = 6;
__host__ __device__ inline int relax2_nchunks(int ni, int RI) {
    const int interior = ni - 2;
    if (interior < 2 * R2_HEAD + RI) return (interior + RI - 1) / RI;   // small block: uniform chunks
    const int nf = (interior - 2 * R2_HEAD) / RI, left = interior - 2 * R2_HEAD - nf * RI;
    return nf + 2 + (left > 0 ? 1 : 0);
}
__host__ __device__ inline void relax2_chunk_rows(int ni, int RI, int rc, int& i0, int& i1) {
    const int interior = ni - 2;
    if (interior < 2 * R2_HEAD + RI) {
        i0 = 1 + rc * RI;
        i1 = (i0 + RI < ni - 1) ? i0 + RI : ni - 1;
        return;
    }
    const int nf = (interior - 2 * R2_HEAD) / RI, left = interior - 2 * R2_HEAD - nf * RI;
    if (rc == 0) {
        i0 = 1;
        i1 = 1 + R2_HEAD;
    } else if (rc <= nf) {
        i0 = 1 + R2_HEAD + (rc - 1) * RI;
        i1 = i0 + RI;
    } else if (left > 0 && rc == nf + 1) {
        i0 = 1 + R2_HEAD + nf * RI;
        i1 = i0 + left;
    } else {
        i0 = ni - 1 - R2_HEAD;
        i1 = ni - 1;
    }
}

// touches a side = reads its perimeter values from `mid` or writes the ring next to it; it matters only where those change (dyn)
__host__ __device__ inline bool relax2_tile_is_border(int ni, int nj, int i0, int i1, int sg, int dyn) {
    const bool top = i0 < 4, bottom = i1 + 1 > ni - 3, left = sg * 240 < 4, right = (sg * 4 + 3) * 60 + 61 > nj - 3;
    return (top && (dyn & 1)) || (bottom && (dyn & 2)) || (left && (dyn & 4)) || (right && (dyn & 8));
}

template <int DOT, int U, int NT, bool W1>
__device__ __forceinline__ void relax2_tile(const Relax2Block& a, int RI, int nSG, int nRC, int subset, int bid) {
    static_assert(U % 3 == 0, "windows rotate by renaming");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int total = nSG * nRC;
    int logical;
    if (subset == R2_BORDER) {   // compact grid: only the border tiles are launched (4000 workgroups that exit at once cost 20 us)
        logical = a.border[bid];
    } else {
        const int q8 = total >> 3, rem = total & 7, xcd = bid & 7, k8 = bid >> 3;
        logical = (xcd < rem) ? xcd * (q8 + 1) + k8 : rem * (q8 + 1) + (xcd - rem) * q8 + k8;
    }
    const int rc = logical / nSG;
    const int sg = logical - rc * nSG;

    const int ni = a.ni, nj = a.nj;
    Relax2Tile t;
    t.c0 = (sg * 4 + wave) * 60;
    t.c = t.c0 - 2 + lane;
    t.out_lane = (lane >= 2) && (lane <= 61) && (t.c >= 1) && (t.c <= nj - 2);
    relax2_chunk_rows(ni, RI, rc, t.i0, t.i1);

    if (subset != R2_ALL && subset != R2_BORDER) {   // workgroup-uniform: the INSIDE launches skip the border tiles
        const bool wg_inside = !relax2_tile_is_border(ni, nj, t.i0, t.i1, sg, a.dyn);
        const bool mine = wg_inside && (subset == R2_INSIDE || ((subset == R2_INSIDE_A) == (rc < nRC / 2)));
        if (!mine) return;   // its partial sums are written by the launch that does run it
    }
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (t.c0 <= nj - 2 && t.i0 < t.i1) {   // wave-uniform
        // strictly inside: columns c0-2 .. c0+61 within [2, nj-3], rows i0-2 .. i1+1 within [2, ni-3] -> no perimeter, no ring
        const bool inside = (t.c0 - 2 >= 2) && (t.c0 + 61 <= nj - 3) && (t.i0 - 2 >= 2) && (t.i1 + 1 <= ni - 3) && ((t.i1 - t.i0) % U == 0);
        if (inside) relax2_strip_inside<DOT, U, NT, W1>(a, t, acc);
        else relax2_strip_edge<DOT, U, NT, W1>(a, t, acc);
    }
    if (DOT != DOT_NONE) block_partials<256, dot_columns(DOT)>(acc, a.partials + static_cast<size_t>(logical) * MAX_PARTIALS);   // slot = tile id in every launch flavour
}

template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256) void k_relax2(Relax2Block a, int RI, int nSG, int nRC, int subset) {
    relax2_tile<DOT, U, NT, W1>(a, RI, nSG, nRC, subset, blockIdx.x);
}

// Ordering against another queue from inside a kernel (Smoother::relax_pairs_pipelined): every workgroup of a WAITED launch spins
// on the counter before it touches memory, then acquires at agent scope (the producer's kernels ended with a release); the first
// thread of a SIGNALLING launch publishes that everything in front of the launch in its queue is complete.
// One thread's wait for `*counter >= target`: relaxed polls with sleeps, bounded in TIME on the constant 100 MHz clock (wall_clock64; a
// poll count would shrink and stretch with the shader clock).  The limit is a hang guard, not an ordering mechanism: whether the two
// queues of a handle can wait for each other at all is settled once, at creation, by Smoother::queue_self_test (limit: milliseconds);
// inside a pass a wait can legitimately last as long as a neighbouring RANK takes to join the exchange in front of it, so the default is
// tens of seconds (QUEUE_WAIT_TICKS).  A wait that runs into its limit raises *error and gives up, and so does every later wait of the
// pass at once (it looks at *error first and every 256 polls) -- the pass drains and returns TM_E_HIP instead of holding the device.
__device__ __forceinline__ void spin_until(const uint32_t* counter, uint32_t target, uint32_t* error, long long limit_ticks) {
    if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const long long t0 = wall_clock64();
    unsigned polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(16);
        ++polls;
        if ((polls & 255u) != 0u) continue;
        if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (wall_clock64() - t0 > limit_ticks) {   // fail the pass rather than hang the device
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
__device__ __forceinline__ void queue_wait_in_kernel(const QueueWait& w) {
    if (threadIdx.x == 0) spin_until(w.counter, w.target, w.error, w.limit_ticks);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void queue_signal_in_kernel(uint32_t* counter) {
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256) void k_relax2_waited(Relax2Block a, int RI, int nSG, int nRC, int subset, QueueWait w) {
    queue_wait_in_kernel(w);
    relax2_tile<DOT, U, NT, W1>(a, RI, nSG, nRC, subset, blockIdx.x);
}

// several blocks of a rank in one launch (see k_apply_batch)
template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256) void k_relax2_batch_waited(Relax2Batch B, int subset, QueueWait w) {
    queue_wait_in_kernel(w);
    int k = 0;
#pragma unroll
    for (int q = 1; q < APPLY_BATCH_MAX; ++q)
        if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
    relax2_tile<DOT, U, NT, W1>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], subset, static_cast<int>(blockIdx.x) - B.start[k]);
}
template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256) void k_relax2_batch(Relax2Batch B, int subset) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < APPLY_BATCH_MAX; ++q)
        if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
    relax2_tile<DOT, U, NT, W1>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], subset, static_cast<int>(blockIdx.x) - B.start[k]);
}

// ------------------------------------------------------------------------------------------
// K2x3: THREE fused Jacobi sweeps per pass, for blocks whose perimeter rows are all `fixed` (a single block with prescribed
// walls: BASELINE configs[1], the slices of configs[4]).  K2x2 is within ~10 % of what the part streams for its traffic
// (tools/ubench/stream.hip) while its fp64 pipes idle a third of the time; a third sweep per pass costs arithmetic only.  With the
// perimeter constant, the perimeter value of every intermediate field is the input's own value at that node, so no `mid` array,
// no ring and no perimeter-row kernel take part.  A wave loads the 64 columns c0-H .. c0+63-H and owns the 64-2H output columns
// c0 .. c0+63-2H (H = 3: 58 columns); row chunks overlap
// by 3 rows either side.  Same arithmetic as three K2 launches, bit for bit (tests/test_gpu_fused.py, test_gpu_benchsize.py).
// ------------------------------------------------------------------------------------------
#ifndef TM_R3_WAVES
#define TM_R3_WAVES 1
#endif
constexpr int R3_U = 3;                 // rows per load group
constexpr int R3_H = 3;                 // halo lanes per side (4 would make every stored segment whole 64 B sectors: slower, 39.4 against 38.4 us per sweep at 4096^2, 12.6 against 11.45 at 2048^2)
constexpr int R3_W = 64 - 2 * R3_H;     // output columns per wave
template <int DOT, int U, int NT, bool W1, bool INSIDE>
__device__ __forceinline__ void relax3_strip(const Relax2Block& a, const Relax2Tile& t, double (&acc)[MAX_PARTIALS]) {
    const int ni = a.ni, nj = a.nj;
    const int cc = INSIDE ? t.c : min(max(t.c, 0), nj - 1);
    const double2* in_col = a.in + cc;
    const int last_row = INSIDE ? t.i1 + 2 : ni - 1;
    auto load_in = [&](int row) { return in_col[static_cast<size_t>(INSIDE ? min(row, last_row) : min(max(row, 0), last_row)) * nj]; };
    const bool perim_col = !INSIDE && ((t.c <= 0) || (t.c >= nj - 1));
    const int nrows = t.i1 - t.i0;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + static_cast<size_t>(t.i0) * nj, 0, nrows * nj * 16, 0x00020000);
    // Coupled blocks (Relax2Block::dyn != 0, Smoother::relax_triples_coupled): the pass treats every perimeter value as frozen, which is
    // wrong along a side whose perimeter rows move -- but three sweeps carry that error only two nodes deep, so the pass simply does
    // not store the nodes within two of such a side; the perimeter-row kernel evaluates them (and the perimeter) level by level.
    const bool skip_col = !INSIDE && (((a.dyn & 4) && t.c <= 2) || ((a.dyn & 8) && t.c >= nj - 3));
    const bool live_lane = t.out_lane && !skip_col;
    const unsigned lane_off = live_lane ? static_cast<unsigned>(cc) * 16u : OOB_VOFFSET;
    const double2 zero = make_double2(0.0, 0.0);

    Row3 A[3], S1[3], S2[3];
    double2 pc[U], pn[U];
    {   // rows i0-3, i0-2 are in the window before the first step (which takes row i0-1)
        const double2 a0 = load_in(t.i0 - 3), a1 = load_in(t.i0 - 2);
#pragma unroll
        for (int u = 0; u < U; ++u) pc[u] = load_in(t.i0 - 1 + u);
        A[0] = make_row(a0);
        A[1] = make_row(a1);
        A[2].c = A[2].e = A[2].h = zero;
        S1[0] = S1[1] = S1[2] = A[2];
        S2[0] = S2[1] = S2[2] = A[2];
    }
    // step s takes row r = i0 - 1 + s of the input, forms s1[r-1], s2[r-2], s3[r-3] and stores row r-3; s3 is live from s = 4 on
    const int nsteps = nrows + 4;
    // (phase 0: the first group, 1: the second, 2: every later one -- a literal at each call, so the tests below fold away)
    auto group = [&](const int tb, const int phase) {
#pragma unroll
        for (int u = 0; u < U; ++u) pn[u] = load_in(t.i0 - 1 + tb + U + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int A0 = u % 3, A1 = (u + 1) % 3, A2 = (u + 2) % 3;   // window slots of rows r-2, r-1, r (and of the same offsets one / two levels up)
            const int r = t.i0 - 1 + tb + u;
            A[A2] = make_row(pc[u]);
            double2 d = zero;
            double2 s1 = relax_row<W1>(A[A0], A[A1], A[A2], a.omega, d);          // row r-1
            if (!INSIDE && (perim_col || r - 1 <= 0 || r - 1 >= ni - 1)) s1 = A[A1].c;   // fixed: the input's own value
            S1[A2] = make_row(s1);
            // the window fills from the top: the second level has its three rows from step 2 on, the third from step 4 on.  The rows
            // they would form before that are never used (and, from windows still holding zeros, would take the slow branch of
            // the reciprocal): skipped -- 6 of the 3 (RI + 4) evaluations of a chunk.
            const bool do2 = phase > 0 || u >= 2, do3 = phase > 1 || (phase == 1 && u >= 1);
            double2 s2 = zero;
            if (do2) {
                s2 = relax_row<W1>(S1[A0], S1[A1], S1[A2], a.omega, d);       // row r-2
                if (!INSIDE && (perim_col || r - 2 <= 0 || r - 2 >= ni - 1)) s2 = S1[A1].c;
            }
            if (do2) S2[A2] = make_row(s2);   // (else: the slot keeps the zeros it was filled with)
            double2 d3 = zero;
            double2 o = zero;
            if (do3) o = relax_row<W1>(S2[A0], S2[A1], S2[A2], a.omega, d3);   // row r-3
            const int i = r - 3;
            const bool row_live = (i >= t.i0) && (i < t.i1) && (INSIDE || !(((a.dyn & 1) && i <= 2) || ((a.dyn & 2) && i >= ni - 3)));   // wave-uniform
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, o), out_rsrc,
                                                   static_cast<int>(row_live ? lane_off + static_cast<unsigned>((i - t.i0) * nj * 16) : OOB_VOFFSET), 0,
                                                   (NT & 1) ? 2 : 0);
            if (DOT == DOT_DELTA) {
                if (!(live_lane && row_live)) d3 = zero;   // masked lanes hold garbage (possibly non-finite): select, never multiply
                accumulate<DOT_DELTA>(acc, S2[A1].c, o, d3);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pc[u] = pn[u];
    };
    static_assert(U == 3, "the phases below assume three steps per group");
    group(0, 0);
    group(U, 1);   // (nsteps >= 5)
    for (int tb = 2 * U; tb < nsteps; tb += U) group(tb, 2);
}

template <int DOT, int U, int NT, bool W1>
__device__ __forceinline__ void relax3_tile(const Relax2Block& a, int RI, int nSG, int nRC, int bid) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int total = nSG * nRC;
    const int q8 = total >> 3, rem = total & 7, xcd = bid & 7, k8 = bid >> 3;
    const int logical = (xcd < rem) ? xcd * (q8 + 1) + k8 : rem * (q8 + 1) + (xcd - rem) * q8 + k8;
    const int rc = logical / nSG;
    const int sg = logical - rc * nSG;
    const int ni = a.ni, nj = a.nj;
    Relax2Tile t;
    t.c0 = (sg * 4 + wave) * R3_W;
    t.c = t.c0 - R3_H + lane;
    t.out_lane = (lane >= R3_H) && (lane < 64 - R3_H) && (t.c >= 1) && (t.c <= nj - 2);
    t.i0 = 1 + rc * RI;
    t.i1 = min(t.i0 + RI, ni - 1);
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (t.c0 <= nj - 2 && t.i0 < t.i1) {   // wave-uniform
        // strictly inside: every column the wave loads within [1, nj-2] and every row it touches within [1, ni-2] -> no fixed value is
        // ever selected, no index is clamped from below
        const bool inside = (t.c0 - R3_H >= 1) && (t.c0 - R3_H + 63 <= nj - 2) && (t.i0 - 3 >= 1) && (t.i1 + 2 <= ni - 2);
        if (inside) relax3_strip<DOT, U, NT, W1, true>(a, t, acc);
        else relax3_strip<DOT, U, NT, W1, false>(a, t, acc);
    }
    if (DOT != DOT_NONE) block_partials<256, dot_columns(DOT)>(acc, a.partials + static_cast<size_t>(logical) * MAX_PARTIALS);
}
template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256, TM_R3_WAVES) void k_relax3(Relax2Block a, int RI, int nSG, int nRC) {
    relax3_tile<DOT, U, NT, W1>(a, RI, nSG, nRC, blockIdx.x);
}
template <int DOT, int U, int NT, bool W1>
__global__ __launch_bounds__(256) void k_relax3_batch(Relax2Batch B) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < APPLY_BATCH_MAX; ++q)
        if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
    relax3_tile<DOT, U, NT, W1>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], static_cast<int>(blockIdx.x) - B.start[k]);
}
static int g_fuse3_rows = 0;   // experiments (TM_FUSE3_ROWS)
// Rows per chunk of K2x3 for the blocks of ONE launch (launch_relax3_blocks batches APPLY_BATCH_MAX blocks).
// The pass is bound by fp64 issue under the power cap, not by bandwidth, so the rows a chunk recomputes for its neighbours
// ((3 RI + 6) / 3 RI of the arithmetic) count: tall chunks, as long as the LAUNCH -- all its blocks together -- still puts ~0.6 x
// 3 workgroups on every CU.  Measured (tools/dev/steady_time.py, us per sweep, two or three boxes each): the height should be
// 2 (mod 12) -- RI + 4 steps are then whole groups of three, and chunk starts 2 x odd rows apart spread over the memory
// channels (4096^2: 38 / 50 rows 37.4 / 36.9-38.2 against 36: 39.5, 44: 37.9-39.7, 56: 41.0, 32 at 2048^2: 11.5; odd heights
// 35 / 39: 40.7 / 41.7).  2048^2: 38 rows 10.25, 26: 10.4 (21 rows, the earlier rule: 11.9); 1448^2: 14 rows 6.7; 1024^2: 8 rows 4.33, 14: 4.67.
void relax3_rows_for_launch(const int* ni, const int* nj, int n, int* rows, bool beside_chain) {
    static const int forced = [] { const char* e = std::getenv("TM_FUSE3_ROWS"); return e ? std::atoi(e) : 0; }();
    if (forced > 0 || g_fuse3_rows > 0) {
        for (int k = 0; k < n; ++k) rows[k] = std::max(1, std::min(forced > 0 ? forced : g_fuse3_rows, ni[k] - 2));
        return;
    }
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 3 * cus;
    }
    // (50 rows: a lone 4096^2 block 36.6 against 37.1 with 38, but 8 x 2048^2 in one launch 75.1 against 73.2-75.3: not offered.)
    // Beside the perimeter-row chain of coupled blocks (relax_triples_coupled) the chain's short kernels queue for the wave slots
    // this pass gives back, once per workgroup lifetime: shorter chunks there (a 2048^2 rank: 38 rows 13.6-13.9, 21-26 rows 13.1).
    static const int heights[] = {38, 26, 14, 8};
    const long need = static_cast<long>(slots) * (beside_chain ? 9 : 6);
    int RI = 8;
    for (const int h : heights) {
        long total = 0;
        for (int k = 0; k < n; ++k) {
            const int nstrips = (nj[k] - 1 + R3_W - 1) / R3_W, nSG = (nstrips + 3) / 4, interior = ni[k] - 2, r = std::max(1, std::min(h, interior));
            total += static_cast<long>(nSG) * ((interior + r - 1) / r);
        }
        RI = h;
        if (total * 10 >= need) break;
    }
    for (int k = 0; k < n; ++k) rows[k] = std::max(1, std::min(RI, ni[k] - 2));
}
int relax3_rows_per_chunk(int ni, int nj) {
    int r = 0;
    relax3_rows_for_launch(&ni, &nj, 1, &r, false);
    return r;
}
bool relax3_supported(int ni, int nj) { return ni >= 7 && nj >= 7 && nj <= (1 << 20); }
int relax3_block_nwg(int ni, int nj, int RI) {
    const int nstrips = (nj - 1 + R3_W - 1) / R3_W;
    return ((nstrips + 3) / 4) * ((ni - 2 + RI - 1) / RI);
}
hipError_t launch_relax3_blocks(const Relax2Block* blocks, const int* rows_per_chunk, int n, int dot, hipStream_t st) {
    for (int first = 0; first < n; first += APPLY_BATCH_MAX) {
        Relax2Batch B;
        B.n = 0;
        int total = 0;
        bool w1 = true;
        const bool nts = blocks[first].store_nt != 0;
        for (int k = first; k < n && B.n < APPLY_BATCH_MAX; ++k) {
            const int q = B.n++;
            B.b[q] = blocks[k];
            B.RI[q] = rows_per_chunk[k];
            B.nSG[q] = ((blocks[k].nj - 1 + R3_W - 1) / R3_W + 3) / 4;
            B.nRC[q] = (blocks[k].ni - 2 + B.RI[q] - 1) / B.RI[q];
            B.start[q] = total;
            total += B.nSG[q] * B.nRC[q];
            w1 = w1 && blocks[k].omega == 1.0;
        }
        for (int q = B.n; q < APPLY_BATCH_MAX; ++q) B.start[q] = total;
        if (total == 0) continue;
        const dim3 grid(total), block(256);
#define TM_R3(KERNEL, ...)                                                                                                   \
    do {                                                                                                                     \
        if (dot == DOT_DELTA) {                                                                                              \
            if (w1 && nts) hipLaunchKernelGGL((KERNEL<DOT_DELTA, R3_U, 1, true>), grid, block, 0, st, __VA_ARGS__);          \
            else if (w1) hipLaunchKernelGGL((KERNEL<DOT_DELTA, R3_U, 0, true>), grid, block, 0, st, __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERNEL<DOT_DELTA, R3_U, 1, false>), grid, block, 0, st, __VA_ARGS__);                   \
        } else {                                                                                                             \
            if (w1 && nts) hipLaunchKernelGGL((KERNEL<DOT_NONE, R3_U, 1, true>), grid, block, 0, st, __VA_ARGS__);           \
            else if (w1) hipLaunchKernelGGL((KERNEL<DOT_NONE, R3_U, 0, true>), grid, block, 0, st, __VA_ARGS__);             \
            else hipLaunchKernelGGL((KERNEL<DOT_NONE, R3_U, 1, false>), grid, block, 0, st, __VA_ARGS__);                    \
        }                                                                                                                    \
    } while (0)
        static const bool force_batch = [] { const char* e = std::getenv("TM_R3_FORCE_BATCH"); return e && std::atoi(e) != 0; }();
        if (B.n == 1 && !force_batch) TM_R3(k_relax3, B.b[0], B.RI[0], B.nSG[0], B.nRC[0]);
        else TM_R3(k_relax3_batch, B);
#undef TM_R3
        const hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// Multigrid: TWO chained applications of a level's frozen-coefficient operator in ONE pass over the level -- the K2x2
// structure (60-of-64 column strips, a 3-row window per stage in registers, rows requested one group ahead, lane / row
// predicates folded into selects and out-of-range buffer-store offsets) applied to the error equation D^-1 A e = f:
//   KIND 0 (POST)  s1 = e + omega (f - D^-1 A e);  out = s1 + omega (f - D^-1 A s1)       two damped-Jacobi sweeps
//                  (PRO: e = e_fine + bilinear interpolation of the coarse correction, formed as the rows enter the window
//                  with k_mg_prolong_add's expression -- the prolonged iterate is never stored)
//   KIND 1 (PRE)   e2 = omega (2 f - omega D^-1 A f)  (the first two sweeps from e = 0, MODE_MG_FIRST2's expression);
//                  out = e2,  out2 = a_ii (f - D^-1 A e2)  (the unscaled residual the restriction averages, MODE_MG_RESID's)
// Each stage is the operation sequence of winslow_row<MODE_SCALED> on the level's frozen field, so the results are bit-identical to
// the two K2 launches they replace (tests/test_gpu_multigrid.py); the metric terms of a node (G11, G22, G12, the reciprocal
// diagonal) are formed ONCE and serve both stages -- the coefficients are frozen -- and the level is read once instead of twice:
// POST 64 B/node instead of 128 (68 instead of 132 with the prolongation), PRE 64 instead of 112.
// The error is zero on the block perimeter (tm_multigrid.hpp): the intermediate stage is forced to zero there.
// ------------------------------------------------------------------------------------------
constexpr int MG_RST_W = 58;   // columns per strip of the PRE pass that also restricts
struct MgMetric {
    double G11, G22, mhG12, m2D, D, rinv;
};
__device__ __forceinline__ MgMetric mg_metric(double2 dxi, double2 det) {
    MgMetric m;
    m.G11 = fma(dxi.x, dxi.x, dxi.y * dxi.y);
    m.G22 = fma(det.x, det.x, det.y * det.y);
    const double G12 = fma(dxi.x, det.x, dxi.y * det.y);
    m.D = m.G11 + m.G22;
    m.m2D = -2.0 * m.D;
    m.mhG12 = -0.5 * G12;
    m.rinv = recip_diag(m.D, m.m2D);
    return m;
}
template <bool HAS_PQ>
__device__ __forceinline__ double2 mg_scaled_row(const MgMetric& g, const Row3& m, const Row3& c, const Row3& p, double P, double Q) {
    double ax = p.c.x + m.c.x, ay = p.c.y + m.c.y;
    double bx = c.h.x, by = c.h.y;
    if (HAS_PQ) {
        const double hP = 0.5 * P, hQ = 0.5 * Q;
        ax = fma(hP, p.c.x - m.c.x, ax);
        ay = fma(hP, p.c.y - m.c.y, ay);
        bx = fma(hQ, c.e.x, bx);
        by = fma(hQ, c.e.y, by);
    }
    const double kx = p.e.x - m.e.x;
    const double ky = p.e.y - m.e.y;
    double sx = g.G22 * ax, sy = g.G22 * ay;
    sx = fma(g.G11, bx, sx);
    sy = fma(g.G11, by, sy);
    sx = fma(g.m2D, c.c.x, sx);
    sy = fma(g.m2D, c.c.y, sy);
    sx = fma(g.mhG12, kx, sx);
    sy = fma(g.mhG12, ky, sy);
    return make_double2(sx * g.rinv, sy * g.rinv);
}

template <int KIND, bool HAS_PQ, bool PRO, bool RST, int U>
__device__ __forceinline__ void mg_pair_strip(const MgPairArgs& a, const Relax2Tile& t) {
    static_assert(!RST || KIND == 1, "the restriction rides behind the residual");
    const int ni = a.ni, nj = a.nj;
    const int cc = min(max(t.c, 0), nj - 1);
    const double2 zero = make_double2(0.0, 0.0);
    auto at = [&](int row) { return static_cast<size_t>(min(max(row, 0), ni - 1)) * nj + cc; };
    const bool perim_col = (t.c <= 0) || (t.c >= nj - 1);
    const bool col_in = (t.c >= 1) && (t.c <= nj - 2);
    const int nrows = t.i1 - t.i0;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + static_cast<size_t>(t.i0) * nj, 0, nrows * nj * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t out2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((KIND == 1 ? a.out2 : a.out) + static_cast<size_t>(t.i0) * nj, 0, nrows * nj * 16, 0x00020000);
    const unsigned lane_off = static_cast<unsigned>(cc) * 16u;
    // PRO: the lane loads the coarse column cj1 of its fine column -- its own when the fine column coincides with a coarse one (even,
    // or an uncoarsened direction), the one to its right when it lies midway (odd); the column cj0 to the left of a midway column is
    // then what the PREVIOUS lane loaded (its column is even: lanes and columns have the same parity, the strip starts at 60 s - 2)
    const int cj1 = PRO ? min(a.cj ? ((cc + 1) >> 1) : cc, a.njc - 1) : 0;
    const bool midway = PRO && a.cj && (cc & 1);

    struct Ld {   // what one step consumes: row i+2 of `in` and of the frozen field, rhs / control function of row i+1
        double2 in, x, f, pq, c0, c1;   // c0, c1: the coarse correction at (ci0, cj1), (ci1, cj1)
    };
    auto load_step = [&](int i, Ld& L) {
        const size_t ar = at(i + 2), aq = at(i + 1);
        L.in = a.in[ar];
        L.x = a.xk[ar];
        if (KIND == 0) L.f = a.f[aq];
        if (HAS_PQ) L.pq = a.pq[aq];
        if (PRO) {
            const int r = min(max(i + 2, 0), ni - 1);
            const int ci0 = min(a.ci ? (r >> 1) : r, a.nic - 1), ci1 = min(a.ci ? ((r + 1) >> 1) : r, a.nic - 1);
            const double2* r0 = a.coarse + static_cast<size_t>(ci0) * a.njc;
            const double2* r1 = a.coarse + static_cast<size_t>(ci1) * a.njc;
            L.c0 = r0[cj1];
            L.c1 = r1[cj1];
        }
    };
    auto entering = [&](int row, const Ld& L) {   // the value of `in` at (row, this lane's column) as stage 1 sees it
        if (!PRO) return L.in;
        // k_mg_prolong_add's expression 0.25 ((a + b) + (c + d)), a = (ci0, cj0), b = (ci0, cj1), c = (ci1, cj0), d = (ci1, cj1);
        // coinciding indices just repeat a value.  Shifts with every lane active, then the selects.
        const double2 l0 = lane_prev0(L.c0), l1 = lane_prev0(L.c1);
        const double2 a0 = midway ? l0 : L.c0, a1 = midway ? l1 : L.c1;
        const double2 v = make_double2(L.in.x + 0.25 * ((a0.x + L.c0.x) + (a1.x + L.c1.x)), L.in.y + 0.25 * ((a0.y + L.c0.y) + (a1.y + L.c1.y)));
        return (col_in && row >= 1 && row <= ni - 2) ? v : L.in;
    };
    auto det_of = [&](double2 x) { return sub2(lane_next0(x), lane_prev0(x)); };

    Row3 A[3], S[3];
    double2 Xc[3], Xe_cur;
    MgMetric Mp = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double2 f_prev = zero, pq_prev = zero;
    double2 rm2 = zero, rm1 = zero;   // RST: the residual of the two rows above
    Ld cur[U], nxt[U];
    {   // warm-up: rows i0-2, i0-1 of `in` and of the frozen field
        Ld w0, w1;
        load_step(t.i0 - 4, w0);
        load_step(t.i0 - 3, w1);
#pragma unroll
        for (int u = 0; u < U; ++u) load_step(t.i0 - 2 + u, cur[u]);
        A[0] = make_row(entering(t.i0 - 2, w0));
        A[1] = make_row(entering(t.i0 - 1, w1));
        Xc[0] = w0.x;
        Xc[1] = w1.x;
        Xe_cur = det_of(w1.x);
        A[2].c = A[2].e = A[2].h = zero;
        S[0] = S[1] = S[2] = A[2];
        Xc[2] = zero;
    }
    const int nsteps = nrows + (RST ? 3 : 2);   // RST: the residual of row i1 too (the last coarse row of the chunk averages it)
    auto group = [&](const int tb) {
#pragma unroll
        for (int u = 0; u < U; ++u) load_step(t.i0 - 2 + tb + U + u, nxt[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int A0 = u % 3, A1 = (u + 1) % 3, A2 = (u + 2) % 3;
            const Ld& L = cur[u];
            const int i = t.i0 - 2 + tb + u;
            const int q = i + 1;
            A[A2] = make_row(entering(i + 2, L));
            Xc[A2] = L.x;
            const double2 Xe_next = det_of(L.x);
            // ---- stage 1 at row q
            const MgMetric M = mg_metric(sub2(Xc[A2], Xc[A0]), Xe_cur);
            const double2 t1 = mg_scaled_row<HAS_PQ>(M, A[A0], A[A1], A[A2], HAS_PQ ? L.pq.x : 0.0, HAS_PQ ? L.pq.y : 0.0);
            const double2 fq = (KIND == 0) ? L.f : A[A1].c;
            double2 s1;
            if (KIND == 0) {
                const double2 res = sub2(fq, t1);
                s1 = make_double2(fma(a.omega, res.x, A[A1].c.x), fma(a.omega, res.y, A[A1].c.y));
            } else {
                s1 = make_double2(a.omega * fma(-a.omega, t1.x, 2.0 * A[A1].c.x), a.omega * fma(-a.omega, t1.y, 2.0 * A[A1].c.y));
            }
            const bool perim_row = (q <= 0) || (q >= ni - 1);
            if (perim_row || perim_col) s1 = zero;   // e = 0 on the block perimeter
            if (KIND == 1) {   // the chunk that owns row q stores e2
                const bool own = t.out_lane && !perim_row && (q >= t.i0) && (q < t.i1);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, s1), out_rsrc,
                                                       static_cast<int>(own ? lane_off + static_cast<unsigned>((q - t.i0) * nj * 16) : OOB_VOFFSET), 0, 0);
            }
            S[A2] = make_row(s1);
            // ---- stage 2 at row i, with the metric formed one step ago
            const double2 t2 = mg_scaled_row<HAS_PQ>(Mp, S[A0], S[A1], S[A2], HAS_PQ ? pq_prev.x : 0.0, HAS_PQ ? pq_prev.y : 0.0);
            const double2 res2 = sub2(f_prev, t2);
            double2 o;
            if (KIND == 0) {
                o = make_double2(fma(a.omega, res2.x, S[A1].c.x), fma(a.omega, res2.y, S[A1].c.y));
            } else {
                const double aii = -0.5 * Mp.D;
                o = make_double2(aii * res2.x, aii * res2.y);
            }
            if (!RST) {
                const bool live = t.out_lane && (i >= t.i0) && (i < t.i1);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, o), out2_rsrc,
                                                       static_cast<int>(live ? lane_off + static_cast<unsigned>((i - t.i0) * nj * 16) : OOB_VOFFSET), 0, 2);
            } else {
                // Full weighting behind the residual (k_mg_restrict's sums in k_mg_restrict's order, so the same bits): the residual
                // is never stored.  Chunks start on odd rows and hold an even number of them, strips start on even columns: the
                // coarse node (ci, cj) sits on the even row 2 ci / even lane of column 2 cj, and its 3 x 3 is complete when the odd
                // row below it leaves stage 2.  The array the unfused kernel reads is zero on the level's perimeter.
                const double2 oz = (col_in && i >= 1 && i <= ni - 2) ? o : zero;
                if ((i & 1) && i > t.i0 && i <= t.i1) {   // wave-uniform
                    double2 acc = zero;
                    auto row_sum = [&](const double2 v, const double wi) {
                        const double2 l = lane_prev0(v), r = lane_next0(v);
                        acc.x = fma(wi * 0.25, l.x, acc.x);
                        acc.y = fma(wi * 0.25, l.y, acc.y);
                        acc.x = fma(wi * 0.5, v.x, acc.x);
                        acc.y = fma(wi * 0.5, v.y, acc.y);
                        acc.x = fma(wi * 0.25, r.x, acc.x);
                        acc.y = fma(wi * 0.25, r.y, acc.y);
                    };
                    row_sum(rm2, 0.25);
                    row_sum(rm1, 0.5);
                    row_sum(oz, 0.25);
                    const int ci = (i - 1) >> 1, cj = t.c >> 1, lane = t.c - t.c0 + 2;
                    const bool emit = !(t.c & 1) && lane >= 4 && lane <= 60 && t.c >= 2 && t.c <= nj - 2 && ci >= 1 && ci <= a.nic - 2 && cj <= a.njc - 2;
                    if (emit) {
                        const size_t oc = static_cast<size_t>(ci) * a.njc + cj;
                        const double2 xp = a.xc[oc + a.njc], xm = a.xc[oc - a.njc], xr = a.xc[oc + 1], xl = a.xc[oc - 1];
                        const double dxx = xp.x - xm.x, dxy = xp.y - xm.y, dex = xr.x - xl.x, dey = xr.y - xl.y;
                        const double aic = -0.5 * (fma(dxx, dxx, dxy * dxy) + fma(dex, dex, dey * dey));
                        const double k = 16.0 / ((aic == 0.0) ? 1.0 : aic);
                        a.fc[oc] = make_double2(k * acc.x, k * acc.y);
                    }
                }
                rm2 = rm1;
                rm1 = oz;
            }
            Mp = M;
            f_prev = fq;
            if (HAS_PQ) pq_prev = L.pq;
            Xe_cur = Xe_next;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    };
    group(0);
    for (int tb = U; tb < nsteps; tb += U) group(tb);
}

template <int KIND, bool HAS_PQ, bool PRO, bool RST = false>
__global__ __launch_bounds__(256) void k_mg_pair(MgPairArgs a, int RI, int nSG, int nRC) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int total = nSG * nRC, bid = blockIdx.x;
    const int q8 = total >> 3, rem = total & 7, xcd = bid & 7, k8 = bid >> 3;   // XCD-aware tile order (see apply_tile)
    const int logical = (xcd < rem) ? xcd * (q8 + 1) + k8 : rem * (q8 + 1) + (xcd - rem) * q8 + k8;
    const int rc = logical / nSG;
    const int sg = logical - rc * nSG;
    Relax2Tile t;
    // RST: 58-column strips on lanes 3..60 -- a coarse node needs the residual of the lanes either side of its own
    t.c0 = (sg * 4 + wave) * (RST ? MG_RST_W : 60);
    t.c = t.c0 - 2 + lane;
    t.out_lane = (lane >= (RST ? 3 : 2)) && (lane <= (RST ? 60 : 61)) && (t.c >= 1) && (t.c <= a.nj - 2);
    t.i0 = 1 + rc * RI;
    t.i1 = min(t.i0 + RI, a.ni - 1);
    if (t.c0 + (RST ? 1 : 0) <= a.nj - 2 && t.i0 < t.i1) mg_pair_strip<KIND, HAS_PQ, PRO, RST, 3>(a, t);   // wave-uniform
}

bool mg_pair_supported(int ni, int nj) { return ni >= 5 && nj >= 5 && nj <= (1 << 20); }
// The restriction folded into the PRE pass: levels coarsened in both directions that fill the device with 18-row chunks (the
// coarser ones are launch-bound either way and keep the two kernels)
bool mg_pair_restrict_supported(int ni, int nj, int ci, int cj) {
    if (!mg_pair_supported(ni, nj) || !ci || !cj) return false;
    const int nstrips = (nj - 2 + MG_RST_W - 1) / MG_RST_W, nSG = (nstrips + 3) / 4;
    return static_cast<long>(nSG) * ((ni - 2 + 17) / 18) >= 1024;
}
hipError_t launch_mg_pair(const MgPairArgs& a, int kind, hipStream_t st) {
    if (!mg_pair_supported(a.ni, a.nj)) return hipErrorInvalidValue;
    if (kind == 1 && a.fc) {
        if (!mg_pair_restrict_supported(a.ni, a.nj, a.ci, a.cj) || !a.xc) return hipErrorInvalidValue;
        const int nstrips = (a.nj - 2 + MG_RST_W - 1) / MG_RST_W, nSG = (nstrips + 3) / 4, RI = 18, nRC = (a.ni - 2 + RI - 1) / RI;
        const dim3 grid(nSG * nRC), block(256);
        if (a.pq) hipLaunchKernelGGL((k_mg_pair<1, true, false, true>), grid, block, 0, st, a, RI, nSG, nRC);
        else hipLaunchKernelGGL((k_mg_pair<1, false, false, true>), grid, block, 0, st, a, RI, nSG, nRC);
        return hipGetLastError();
    }
    const int nstrips = (a.nj - 1 + 59) / 60, nSG = (nstrips + 3) / 4, interior = a.ni - 2;
    int RI = 18;   // like K2x2: short chunks, and shorter still while the level cannot fill the device
    while (RI > 3 && static_cast<long>(nSG) * ((interior + RI - 1) / RI) < 1024) RI -= 3;
    RI = std::max(1, std::min(RI, interior));
    const int nRC = (interior + RI - 1) / RI;
    const dim3 grid(nSG * nRC), block(256);
    const bool pq = a.pq != nullptr, pro = a.coarse != nullptr;
    if (kind == 1) {
        if (pq) hipLaunchKernelGGL((k_mg_pair<1, true, false>), grid, block, 0, st, a, RI, nSG, nRC);
        else hipLaunchKernelGGL((k_mg_pair<1, false, false>), grid, block, 0, st, a, RI, nSG, nRC);
    } else if (pro) {
        if (pq) hipLaunchKernelGGL((k_mg_pair<0, true, true>), grid, block, 0, st, a, RI, nSG, nRC);
        else hipLaunchKernelGGL((k_mg_pair<0, false, true>), grid, block, 0, st, a, RI, nSG, nRC);
    } else {
        if (pq) hipLaunchKernelGGL((k_mg_pair<0, true, false>), grid, block, 0, st, a, RI, nSG, nRC);
        else hipLaunchKernelGGL((k_mg_pair<0, false, false>), grid, block, 0, st, a, RI, nSG, nRC);
    }
    return hipGetLastError();
}

static int g_fuse_rows = 0;   // 0 = choose per block (relax2_rows_per_chunk); > 0 = forced (tm_tune_fuse)
constexpr int R2_U = 3;      // rows per load group of k_relax2

// Rows per chunk of K2x2.  Short chunks win on the MI355X although every chunk re-reads 4 rows: with several times more
// workgroups than the 4 x #CU that fit at once, their load-heavy prologues and arithmetic-heavy steady states interleave in
// time instead of marching in lockstep (measured at 4096^2: 18 rows 110 us, 36 rows 116 us, 75 rows = one round 125 us,
// 150 rows 138 us).  18 rows, fewer when the block is too small to fill the device even then.
int relax2_rows_per_chunk(int ni, int nj) {
    const int interior = ni - 2;
    if (g_fuse_rows > 0) return std::max(1, std::min(g_fuse_rows, interior));
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 4 * cus;
    }
    const int nstrips = (nj - 1 + 59) / 60, nSG = (nstrips + 3) / 4;
    int RI = 6 * R2_U;
    while (RI > 4 * R2_U && static_cast<long>(nSG) * ((interior + RI - 1) / RI) < slots) RI -= R2_U;
    return std::max(1, std::min(RI, interior));
}
bool relax2_supported(int ni, int nj) { return ni >= 5 && nj >= 5 && nj <= (1 << 20); }   // buffer-store offsets: rows_per_chunk * nj * 16 < 2^31
int relax2_block_nwg(int ni, int nj, int RI) {
    const int nstrips = (nj - 1 + 59) / 60;
    return ((nstrips + 3) / 4) * relax2_nchunks(ni, RI);
}
std::vector<int32_t> relax2_border_tiles(int ni, int nj, int RI, int dyn) {
    const int nstrips = (nj - 1 + 59) / 60, nSG = (nstrips + 3) / 4, nRC = relax2_nchunks(ni, RI);
    std::vector<int32_t> ids;
    for (int rc = 0; rc < nRC; ++rc) {
        int i0, i1;
        relax2_chunk_rows(ni, RI, rc, i0, i1);
        for (int sg = 0; sg < nSG; ++sg)
            if (relax2_tile_is_border(ni, nj, i0, i1, sg, dyn)) ids.push_back(rc * nSG + sg);
    }
    return ids;
}
// Launch dispatch over (partial sums, omega == 1, cache policy).  NTBITS: bit 0 = streaming (nt) result stores, bit 1 = streaming loads.
// Result stores: streaming when the rank's fields are far larger than the 256 MB Infinity Cache or small enough for the L2s
// (Relax2Block::store_nt, chosen by the handle from its footprint), plain in between -- a pass then finds the field the previous
// pass wrote still in the Infinity Cache (lone 2048^2 block: 16.5 -> 14.2 us per sweep; 4096^2: streaming 111 us against 120 us per
// pass; tools/dev/ab_saux.sh, tools/ubench/stream.hip).  Streaming loads: the capped interior pass of a multi-rank pair (lds > 0).
#define TM_R2_DISPATCH(KERNEL, NTBITS, ...)                                                                                        \
    do {                                                                                                                           \
        if (dot == DOT_DELTA) {                                                                                                    \
            if (w1) hipLaunchKernelGGL((KERNEL<DOT_DELTA, R2_U, NTBITS, true>), grid, block, lds, st, __VA_ARGS__);                 \
            else hipLaunchKernelGGL((KERNEL<DOT_DELTA, R2_U, (NTBITS) & 1, false>), grid, block, lds, st, __VA_ARGS__);             \
        } else {                                                                                                                   \
            if (w1) hipLaunchKernelGGL((KERNEL<DOT_NONE, R2_U, NTBITS, true>), grid, block, lds, st, __VA_ARGS__);                  \
            else hipLaunchKernelGGL((KERNEL<DOT_NONE, R2_U, (NTBITS) & 1, false>), grid, block, lds, st, __VA_ARGS__);              \
        }                                                                                                                          \
    } while (0)
hipError_t launch_relax2_block(const Relax2Block& a, int RI, int dot, int subset, hipStream_t st, size_t lds, const QueueWait* wait) {
    const int nstrips = (a.nj - 1 + 59) / 60;
    const int nSG = (nstrips + 3) / 4, nRC = relax2_nchunks(a.ni, RI);
    if (subset == R2_BORDER && a.nborder == 0) return wait ? launch_queue_wait(wait->counter, wait->target, wait->error, st, wait->limit_ticks) : hipSuccess;
    const dim3 grid(subset == R2_BORDER ? a.nborder : nSG * nRC), block(256);
    const bool w1 = a.omega == 1.0;
    const bool nts = a.store_nt != 0;
    if (wait) {
        const QueueWait w = *wait;
        if (nts) TM_R2_DISPATCH(k_relax2_waited, 1, a, RI, nSG, nRC, subset, w);
        else TM_R2_DISPATCH(k_relax2_waited, 0, a, RI, nSG, nRC, subset, w);
        return hipGetLastError();
    }
    // lds > 0 = the capped interior pass of a multi-rank sweep pair on a small block: streaming loads (relax2_strip_inside)
    static const int inside_ntl = [] { const char* e = std::getenv("TM_R2_INSIDE_NTL"); return e ? std::atoi(e) : -1; }();   // experiment knob
    if (lds > 0 && w1 && (inside_ntl < 0 || inside_ntl == 1)) {
        if (nts) TM_R2_DISPATCH(k_relax2, 3, a, RI, nSG, nRC, subset);
        else TM_R2_DISPATCH(k_relax2, 2, a, RI, nSG, nRC, subset);
        return hipGetLastError();
    }
    if (nts) TM_R2_DISPATCH(k_relax2, 1, a, RI, nSG, nRC, subset);
    else TM_R2_DISPATCH(k_relax2, 0, a, RI, nSG, nRC, subset);
    return hipGetLastError();
}
hipError_t launch_relax2_blocks(const Relax2Block* blocks, const int* rows_per_chunk, int n, int dot, int subset, hipStream_t st, size_t lds, const QueueWait* wait) {
    if (n == 1) return launch_relax2_block(blocks[0], rows_per_chunk[0], dot, subset, st, lds, wait);
    bool waited = false;   // the first launch of the group carries the wait (in-order queue: the later ones follow it)
    for (int first = 0; first < n; first += APPLY_BATCH_MAX) {
        Relax2Batch B;
        B.n = 0;
        int total = 0;
        bool w1 = true;
        const bool nts = blocks[first].store_nt != 0;   // one policy per handle
        for (int k = first; k < n && B.n < APPLY_BATCH_MAX; ++k) {
            const int q = B.n++;
            B.b[q] = blocks[k];
            B.RI[q] = rows_per_chunk[k];
            B.nSG[q] = ((blocks[k].nj - 1 + 59) / 60 + 3) / 4;
            B.nRC[q] = relax2_nchunks(blocks[k].ni, B.RI[q]);
            B.start[q] = total;
            total += (subset == R2_BORDER) ? blocks[k].nborder : B.nSG[q] * B.nRC[q];
            w1 = w1 && blocks[k].omega == 1.0;
        }
        for (int q = B.n; q < APPLY_BATCH_MAX; ++q) B.start[q] = total;
        if (total == 0) continue;
        const dim3 grid(total), block(256);
        if (wait && !waited) {
            waited = true;
            const QueueWait w = *wait;
            if (nts) TM_R2_DISPATCH(k_relax2_batch_waited, 1, B, subset, w);
            else TM_R2_DISPATCH(k_relax2_batch_waited, 0, B, subset, w);
        } else if (lds > 0 && w1) {   // see launch_relax2_block
            if (nts) TM_R2_DISPATCH(k_relax2_batch, 3, B, subset);
            else TM_R2_DISPATCH(k_relax2_batch, 2, B, subset);
        } else {
            if (nts) TM_R2_DISPATCH(k_relax2_batch, 1, B, subset);
            else TM_R2_DISPATCH(k_relax2_batch, 0, B, subset);
        }
        const hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
    }
    if (wait && !waited) return launch_queue_wait(wait->counter, wait->target, wait->error, st, wait->limit_ticks);   // nothing to launch: the wait alone
    return hipSuccess;
}
#undef TM_R2_DISPATCH
void tune_fuse_rows(int rows) { g_fuse_rows = rows > 0 ? rows : 0; }

// Rows per chunk of K2.  A block that cannot fill the device is bound by the chain of dependent row-group loads in each wave
// (a round trip per 3 rows), not by bandwidth: it gets the shortest chunks (multiples of the 3-row load group) that keep the launch
// within 512 workgroups = two per CU (T106, 8 blocks of 10^2..10^4 nodes: 42.5 -> 29.7 us per BiCGStab iteration with 3 rows
// instead of 18; a 1024^2 block, 9 rows: 88 -> 81 us; with 1024 workgroups the lazy scalar steps no longer apply: 95 us).  `rows` > 0: the caller's choice (a handle that looked at all of its blocks, Smoother::create).
static inline int rows_per_chunk(int ni, int nj, int rows) {
    const int interior = ni - 2;
    int RI = rows > 0 ? rows : g_rows_per_chunk;
    if (rows <= 0 && !g_rows_forced) {
        const int nSG = (nj + 255) / 256;
        int r = 3;
        while (r < RI && nSG * ((interior + r - 1) / r) > 512) r += 3;
        RI = std::min(RI, r);
    }
    if (RI > interior) RI = interior;
    if (RI < 1) RI = 1;
    return RI;
}
int apply_block_nwg(int ni, int nj, int rows) {
    const int RI = rows_per_chunk(ni, nj, rows);
    const int nSG = (nj + 255) / 256;
    const int nRC = (ni - 2 + RI - 1) / RI;
    return nSG * nRC;
}
// strip groups (4 waves each) of the overlapping-strip layout: 62 owned columns per wave over the interior columns 1 .. nj-2
static inline int overlap_strip_groups(int nj) { return ((nj - 2 + 61) / 62 + 3) / 4; }
int apply_block_nwg_overlap(int ni, int nj, int rows) {
    const int RI = rows_per_chunk(ni, nj, rows);
    const int nRC = (ni - 2 + RI - 1) / RI;
    return overlap_strip_groups(nj) * nRC;
}

template <int MODE, int DOT, bool FIELD, bool HAS_PQ>
static hipError_t launch_apply_u(const ApplyBlock& a, int RI, int nSG, int nRC, hipStream_t st) {
    const dim3 grid(nSG * nRC), block(256);
#define TM_K2(U_, N_) hipLaunchKernelGGL((k_apply<MODE, DOT, FIELD, HAS_PQ, U_, N_>), grid, block, 0, st, a, RI, nSG, nRC)
    if (g_unroll >= 6 && RI >= 6) {
        if (g_nt) TM_K2(6, true); else TM_K2(6, false);
    } else {
        if (g_nt) TM_K2(3, true); else TM_K2(3, false);
    }
#undef TM_K2
    return hipGetLastError();
}
template <int MODE, int DOT>
static hipError_t launch_apply_md(const ApplyBlock& a, int RI, int nSG, int nRC, hipStream_t st) {
    const bool field = (a.in == a.xk);
    const bool pq = a.pq != nullptr;
    if (field && pq) return launch_apply_u<MODE, DOT, true, true>(a, RI, nSG, nRC, st);
    if (field) return launch_apply_u<MODE, DOT, true, false>(a, RI, nSG, nRC, st);
    if (pq) return launch_apply_u<MODE, DOT, false, true>(a, RI, nSG, nRC, st);
    return launch_apply_u<MODE, DOT, false, false>(a, RI, nSG, nRC, st);
}

template <int MODE, int DOT, bool FIELD, bool HAS_PQ>
static hipError_t launch_batch_u(const ApplyBatch& B, int total, hipStream_t st) {
    const dim3 grid(total), block(256);
    if (g_nt) hipLaunchKernelGGL((k_apply_batch<MODE, DOT, FIELD, HAS_PQ, 3, true>), grid, block, 0, st, B);
    else hipLaunchKernelGGL((k_apply_batch<MODE, DOT, FIELD, HAS_PQ, 3, false>), grid, block, 0, st, B);
    return hipGetLastError();
}
template <int MODE, int DOT>
static hipError_t launch_batch_md(const ApplyBatch& B, int total, bool field, bool pq, hipStream_t st) {
    if (field && pq) return launch_batch_u<MODE, DOT, true, true>(B, total, st);
    if (field) return launch_batch_u<MODE, DOT, true, false>(B, total, st);
    if (pq) return launch_batch_u<MODE, DOT, false, true>(B, total, st);
    return launch_batch_u<MODE, DOT, false, false>(B, total, st);
}

// Several blocks, one launch per group of APPLY_BATCH_MAX.  All blocks share mode / dot and the field-mode / control-function
// flavour; each block's `partials` already points at its own slots.  Modes: the ones the Krylov and relaxation loops use.
hipError_t launch_apply_blocks(const ApplyBlock* blocks, int n, int mode, int dot, hipStream_t st) {
    if (n == 1) return launch_apply_block(blocks[0], mode, dot, st);
    for (int first = 0; first < n; first += APPLY_BATCH_MAX) {
        ApplyBatch B;
        B.n = 0;
        int total = 0;
        bool field = true, pq = true;
        for (int k = first; k < n && B.n < APPLY_BATCH_MAX; ++k) {
            const ApplyBlock& a = blocks[k];
            if (a.ni < 3 || a.nj < 3) continue;   // no interior rows
            const int q = B.n++;
            B.b[q] = a;
            B.RI[q] = rows_per_chunk(a.ni, a.nj, a.rows);
            B.nSG[q] = (a.nj + 255) / 256;
            B.nRC[q] = (a.ni - 2 + B.RI[q] - 1) / B.RI[q];
            B.start[q] = total;
            total += B.nSG[q] * B.nRC[q];
            field = field && (a.in == a.xk);
            pq = pq && (a.pq != nullptr);
        }
        if (B.n == 0) continue;
        for (int q = B.n; q < APPLY_BATCH_MAX; ++q) B.start[q] = total;
        hipError_t e = hipErrorInvalidValue;
        if (mode == MODE_SCALED && dot == DOT_NONE) e = launch_batch_md<MODE_SCALED, DOT_NONE>(B, total, field, pq, st);
        else if (mode == MODE_RAW && dot == DOT_NONE) e = launch_batch_md<MODE_RAW, DOT_NONE>(B, total, field, pq, st);
        else if (mode == MODE_SCALED && dot == DOT_AUX) e = launch_batch_md<MODE_SCALED, DOT_AUX>(B, total, field, pq, st);
        else if (mode == MODE_SCALED && dot == DOT_AUX2) e = launch_batch_md<MODE_SCALED, DOT_AUX2>(B, total, field, pq, st);
        else if (mode == MODE_SCALED && dot == DOT_IN) e = launch_batch_md<MODE_SCALED, DOT_IN>(B, total, field, pq, st);
        else if (mode == MODE_RESID && dot == DOT_OUT2) e = launch_batch_md<MODE_RESID, DOT_OUT2>(B, total, field, pq, st);
        else if (mode == MODE_RESID && dot == DOT_NONE) e = launch_batch_md<MODE_RESID, DOT_NONE>(B, total, field, pq, st);
        else if (mode == MODE_RELAX && dot == DOT_DELTA) e = launch_batch_md<MODE_RELAX, DOT_DELTA>(B, total, field, pq, st);
        else if (mode == MODE_RELAX && dot == DOT_NONE) e = launch_batch_md<MODE_RELAX, DOT_NONE>(B, total, field, pq, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_apply_block(const ApplyBlock& a, int mode, int dot, hipStream_t st) {
    if (a.ni < 3 || a.nj < 3) return hipSuccess;   // no interior rows
    const int RI = rows_per_chunk(a.ni, a.nj, a.rows);
    const int nSG = (a.nj + 255) / 256;
    const int nRC = (a.ni - 2 + RI - 1) / RI;
    // the (mode, dot) pairs the smoother uses
    if (mode == MODE_RAW && dot == DOT_NONE) return launch_apply_md<MODE_RAW, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_SCALED && dot == DOT_NONE) return launch_apply_md<MODE_SCALED, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_SCALED && dot == DOT_AUX) return launch_apply_md<MODE_SCALED, DOT_AUX>(a, RI, nSG, nRC, st);
    if (mode == MODE_SCALED && dot == DOT_AUX2) return launch_apply_md<MODE_SCALED, DOT_AUX2>(a, RI, nSG, nRC, st);
    if (mode == MODE_SCALED && dot == DOT_IN) return launch_apply_md<MODE_SCALED, DOT_IN>(a, RI, nSG, nRC, st);
    if (mode == MODE_RESID && dot == DOT_OUT2) return launch_apply_md<MODE_RESID, DOT_OUT2>(a, RI, nSG, nRC, st);
    if (mode == MODE_RESID && dot == DOT_NONE) return launch_apply_md<MODE_RESID, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_RELAX && dot == DOT_DELTA) return launch_apply_md<MODE_RELAX, DOT_DELTA>(a, RI, nSG, nRC, st);
    if (mode == MODE_RELAX && dot == DOT_NONE) return launch_apply_md<MODE_RELAX, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_MG_FIRST2) {
        if (a.pq) return launch_apply_u<MODE_MG_FIRST2, DOT_NONE, false, true>(a, RI, nSG, nRC, st);
        return launch_apply_u<MODE_MG_FIRST2, DOT_NONE, false, false>(a, RI, nSG, nRC, st);
    }
    if (mode == MODE_MG_RESID || mode == MODE_MG_SMOOTH) {   // never field mode: the frozen coordinates are a different array
        const bool smooth = mode == MODE_MG_SMOOTH;
        if (a.pq) return smooth ? launch_apply_u<MODE_MG_SMOOTH, DOT_NONE, false, true>(a, RI, nSG, nRC, st) : launch_apply_u<MODE_MG_RESID, DOT_NONE, false, true>(a, RI, nSG, nRC, st);
        return smooth ? launch_apply_u<MODE_MG_SMOOTH, DOT_NONE, false, false>(a, RI, nSG, nRC, st) : launch_apply_u<MODE_MG_RESID, DOT_NONE, false, false>(a, RI, nSG, nRC, st);
    }
    if (mode == MODE_DIAG_COPY) return launch_apply_md<MODE_DIAG_COPY, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_DIAG_SUM9) return launch_apply_md<MODE_DIAG_SUM9, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_DIAG_NOSTORE) return launch_apply_md<MODE_DIAG_NOSTORE, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_DIAG_NOLOAD) return launch_apply_md<MODE_DIAG_NOLOAD, DOT_NONE>(a, RI, nSG, nRC, st);
    if (mode == MODE_DIAG_MATH) return launch_apply_md<MODE_DIAG_MATH, DOT_NONE>(a, RI, nSG, nRC, st);
    return hipErrorInvalidValue;
}

void tune_apply(int rows, int unroll, int pipe, int nt) {
    if (rows > 0) {
        g_rows_per_chunk = rows;
        g_rows_forced = true;
    }
    if (unroll > 0) g_unroll = unroll;
    (void)pipe;
    if (nt >= 0) g_nt = nt;
}

// ------------------------------------------------------------------------------------------
// K4/K5 perimeter rows: fixed / connected / junction / sliding rows (static coefficients,
// smooth.zig:780-921, 1115-1165) and `smoothed` interface rows (cross-block 9-point stencil,
// smooth.zig:994-1105).  One thread per row; columns are visited in ascending global id
// (the reference's CSR order) so the row sum matches the CPU mat-vec bit for bit.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double pick9(const double (&c)[9], int s) {
    double v = c[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) v = (s == k) ? c[k] : v;
    return v;
}

// Point k of run R.  fetch_in(q, id) / fetch_xk(m, id) deliver column q (metric neighbour m) whose local vector index is id,
// fetch_self(id) the row's own value: plain loads in the perimeter-row kernel; the border workgroups of K2x2, which evaluate the
// perimeter rows of their own columns in place (FoldRun), take the first-interior ring from registers instead.  Leaves the
// thread's contributions to the fused dot products in acc[].
template <int MODE, int DOT, class FIn, class FXk, class FSelf>
__device__ __forceinline__ void edge_row_eval(const EdgeRun& R, const double* __restrict__ rhs, int k, FIn fetch_in, FXk fetch_xk, FSelf fetch_self,
                                              const double2* __restrict__ pq, const double2* __restrict__ aux, double omega, double (&acc)[MAX_PARTIALS],
                                              double2& result, int& row_out_id) {
    const int row = R.row0 + k * R.row_stride;
    row_out_id = row;
    const int kind = R.kind;
    const int nc = R.ncols;
    const int self = R.self;
    auto col = [&](int q) { return R.col0[q] + k * R.col_stride[q]; };
    double sx = 0.0, sy = 0.0, rhs_x, rhs_y, diag_x, diag_y;
    if (MODE == MODE_RELAX && kind == 5) {
        // interior node of a REMOTE block, evaluated here as a ghost row (depth-2 halo): K2's own arithmetic on the gathered
        // 3 x 3 neighbourhood (columns in (i-1,j-1) ... (i+1,j+1) order), bit-identical to what the owner's K2 / K2x2 stores
        const double2 ml = fetch_in(0, col(0)), mc = fetch_in(1, col(1)), mr = fetch_in(2, col(2));
        const double2 cl = fetch_in(3, col(3)), cc = fetch_in(4, col(4)), cr = fetch_in(5, col(5));
        const double2 pl = fetch_in(6, col(6)), pc = fetch_in(7, col(7)), pr = fetch_in(8, col(8));
        const double2 c_e = sub2(cr, cl);
        double2 delta;
        result = winslow_row<MODE_RELAX, false>(mc, sub2(mr, ml), cc, c_e, add2(cr, cl), pc, sub2(pr, pl), sub2(pc, mc), c_e, 0.0, 0.0, omega, delta);
        if (DOT == DOT_DELTA && (R.flags & 16)) accumulate<DOT_DELTA>(acc, cc, result, delta);   // an OWN interior node (coupled triples): its displacement counts
        return;
    }
    if (kind == 1 /* smoothed */) {
        const double2 im1_j = fetch_xk(0, R.met0[0] + k * R.met_stride[0]), ip1_j = fetch_xk(1, R.met0[1] + k * R.met_stride[1]);
        const double2 i_jm1 = fetch_xk(2, R.met0[2] + k * R.met_stride[2]);
        double2 i_jp1 = fetch_xk(3, R.met0[3] + k * R.met_stride[3]);
        const bool periodic = R.flags & 1;
        const double per_x = R.per[0], per_y = R.per[1];
        if (periodic) {   // types.add(p, types.neg(periodicity)), smooth.zig:1032
            i_jp1.x = i_jp1.x + (-per_x);
            i_jp1.y = i_jp1.y + (-per_y);
        }
        const double2 cf = pq ? pq[row] : make_double2(0.0, 0.0);
        // periodic rows pass (P,Q), non-periodic rows pass (Q,P): smooth.zig:1040-1041 vs 1082-1083
        const double P = periodic ? cf.x : cf.y, Q = periodic ? cf.y : cf.x;
        double c[9];
        stencil_coefs<true>(im1_j, ip1_j, i_jm1, i_jp1, P, Q, c);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double ck = pick9(c, R.slot[q]);
            const double2 w = fetch_in(q, col(q));
            sx += ck * w.x;
            sy += ck * w.y;
        }
        diag_x = diag_y = c[S_I_J];
        if (periodic) {   // smooth.zig:1060-1061
            const double cs = c[S_IM1_JP1] + c[S_I_JP1] + c[S_IP1_JP1];
            rhs_x = per_x * cs;
            rhs_y = per_y * cs;
        } else {
            rhs_x = 0.0;
            rhs_y = 0.0;
        }
    } else {
        diag_x = 0.0;
        diag_y = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (q < nc) {
                const double ax = R.cx[q], ay = R.cy[q];
                const double2 w = fetch_in(q, col(q));
                sx += ax * w.x;
                sy += ay * w.y;
                if (q == self) {
                    diag_x = ax;
                    diag_y = ay;
                }
            }
        }
        rhs_x = rhs[2 * (R.first + k)];
        rhs_y = rhs[2 * (R.first + k) + 1];
    }
    const double2 w_self = fetch_self(row);
    // ghost copies of rows whose right-hand side is the node's own boundary coordinate (fixed rows, the x-system of sliding
    // rows): that coordinate IS the row's current value -- such a row reproduces it in every sweep
    if (R.flags & 4) rhs_x = w_self.x;
    if (R.flags & 8) rhs_y = w_self.y;
    // constraint rows are enforced exactly in a relaxation sweep (omega = 1); smoothed rows relax like interior rows
    const double om = (kind == 1) ? omega : 1.0;
    double2 o;
    o.x = row_out<MODE>(sx, rhs_x, diag_x, w_self.x, om);
    o.y = row_out<MODE>(sy, rhs_y, diag_y, w_self.y, om);
    result = o;
    accumulate<DOT>(acc, w_self, o, (DOT == DOT_AUX || DOT == DOT_AUX2 || DOT == DOT_B2) ? aux[row] : ((DOT == DOT_DELTA) ? make_double2(o.x - w_self.x, o.y - w_self.y) : o));
}

// Workgroup `wg` of a perimeter-row pass; `tid` = thread within it (threads >= EDGE_BLOCK of a wider block idle).
// The vectors and scalars of a virtual-input pass (VirtualIn + the scalars read from the scalar block), as the device code sees them
struct VirtualArgs {
    const double2 *in2 = nullptr, *in3 = nullptr, *in4 = nullptr;
    double2 *pout = nullptr, *rout = nullptr, *uio = nullptr;
    double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0), vc = make_double2(0.0, 0.0);
};

template <int MODE, int DOT, int VK = VK_NONE>
__device__ __forceinline__ void edge_rows_wg(const EdgeRowsDev& e, int wg, int tid, const double2* __restrict__ in, const double2* __restrict__ xk,
                                             const double2* __restrict__ pq, const double2* __restrict__ aux, double2* __restrict__ out, double omega,
                                             double (&acc)[MAX_PARTIALS], const VirtualArgs& V = VirtualArgs()) {
    // one workgroup = one stretch of one run: everything read through R is workgroup-uniform (scalar loads)
    const EdgeRun& R = e.runs[__builtin_amdgcn_readfirstlane(e.wg_run[wg])];
    const int k = __builtin_amdgcn_readfirstlane(e.wg_k0[wg]) + tid;
    if (tid < EDGE_BLOCK && k < R.count) {
        double2 o;
        int row;
        // VK: the vector is formed on the fly (virtual_vec / virtual_r); VK_P and VK_R also store it for the row's own node
        auto vec = [&](int id) {
            const double2 x = in[id];
            if (VK == VK_NONE) return x;
            const double2 y = V.in2[id];
            if (VK == VK_R) return virtual_r(x, y, V.in3[id], V.in4[id], V.va, V.vb, V.vc).pn;
            const double2 z = VK == VK_P ? V.in3[id] : y;
            return virtual_vec<VK>(x, y, z, V.va, V.vb);
        };
        edge_row_eval<MODE, DOT>(R, e.rhs, k, [&](int, int id) { return vec(id); }, [&](int, int id) { return xk[id]; }, [&](int id) { return vec(id); }, pq, aux,
                                 omega, acc, o, row);
        out[row] = o;
        if (VK == VK_P) V.pout[row] = vec(row);
        if (VK == VK_R) {   // the node's own share of the x / r / p updates (k_xr_update_vs's expressions)
            const double2 w = V.in4[row];
            const VirtualR q = virtual_r(in[row], V.in2[row], V.in3[row], w, V.va, V.vb, V.vc);
            double2 un = V.uio[row];
            un.x += V.va.x * w.x;
            un.y += V.va.y * w.y;
            un.x += V.vb.x * q.s.x;
            un.y += V.vb.y * q.s.y;
            V.uio[row] = un;
            V.rout[row] = q.rn;
            V.pout[row] = q.pn;
            acc[2] += q.rn.x * q.rn.x;
            acc[3] += q.rn.y * q.rn.y;
        }
    }
}

template <int MODE, int DOT>
__global__ __launch_bounds__(EDGE_BLOCK) void k_edge_rows(EdgeRowsDev e, const double2* __restrict__ in,
                                                          const double2* __restrict__ xk, const double2* __restrict__ pq,
                                                          const double2* __restrict__ aux, double2* __restrict__ out, double omega,
                                                          double* partials, uint32_t* signal) {
    queue_signal_in_kernel(signal);
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    edge_rows_wg<MODE, DOT>(e, blockIdx.x, threadIdx.x, in, xk, pq, aux, out, omega, acc);
    if (DOT != DOT_NONE) block_partials<EDGE_BLOCK, dot_columns(DOT)>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}

// The three level passes of a coupled sweep triple in one launch (FusedLevelsDev, tm_kernels.h): strip `blockIdx.x`, levels 1..3 with a
// barrier in between; a wave takes whole tasks, so the run descriptor is wave-uniform as in k_edge_rows.  Same edge_row_eval, same bits.
template <int DOT>
__global__ __launch_bounds__(LEVELS_BLOCK) void k_edge_levels3(FusedLevelsDev F, EdgeRowsDev e1, EdgeRowsDev e2, EdgeRowsDev e3, const double2* x, double2* m,
                                                               double2* m2, double2* out, const double2* __restrict__ pq, double omega, double* partials) {
    const int s = blockIdx.x;
    const int wave = static_cast<int>(threadIdx.x) >> 6, lane = static_cast<int>(threadIdx.x) & 63;
    constexpr int NW = LEVELS_BLOCK / 64;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    auto level = [&](const EdgeRowsDev& e, int l, const double2* in, double2* dst, auto dot_tag) {
        constexpr int D = decltype(dot_tag)::value;
        const int t0 = F.off[4 * s + l], t1 = F.off[4 * s + l + 1];
        for (int t = t0 + wave; t < t1; t += NW) {
            const int run = __builtin_amdgcn_readfirstlane(F.tasks[t].run);
            const int k0 = __builtin_amdgcn_readfirstlane(F.tasks[t].k0), cnt = __builtin_amdgcn_readfirstlane(F.tasks[t].count);
            const EdgeRun& R = e.runs[run];
            if (lane < cnt) {
                double2 o;
                int row;
                edge_row_eval<MODE_RELAX, D>(R, e.rhs, k0 + lane, [&](int, int id) { return in[id]; }, [&](int, int id) { return in[id]; },
                                             [&](int id) { return in[id]; }, pq, nullptr, omega, acc, o, row);
                dst[row] = o;
            }
        }
    };
    level(e1, 0, x, m, std::integral_constant<int, DOT_NONE>{});
    __syncthreads();   // workgroup-scope release / acquire: level 2 reads what this workgroup stored at level 1
    level(e2, 1, m, m2, std::integral_constant<int, DOT_NONE>{});
    __syncthreads();
    level(e3, 2, m2, out, std::integral_constant<int, DOT>{});
    if (DOT != DOT_NONE) block_partials<LEVELS_BLOCK, dot_columns(DOT)>(acc, partials + static_cast<size_t>(s) * MAX_PARTIALS);
}
hipError_t launch_edge_levels3(const FusedLevelsDev& F, const EdgeRowsDev& e1, const EdgeRowsDev& e2, const EdgeRowsDev& e3, const double2* x, double2* m,
                               double2* m2, double2* out, const double2* pq, double omega, int dot, double* partials, hipStream_t st) {
    if (F.nstrips == 0) return hipSuccess;
    if (dot == DOT_DELTA) hipLaunchKernelGGL(k_edge_levels3<DOT_DELTA>, dim3(F.nstrips), dim3(LEVELS_BLOCK), 0, st, F, e1, e2, e3, x, m, m2, out, pq, omega, partials);
    else if (dot == DOT_NONE) hipLaunchKernelGGL(k_edge_levels3<DOT_NONE>, dim3(F.nstrips), dim3(LEVELS_BLOCK), 0, st, F, e1, e2, e3, x, m, m2, out, pq, omega, partials);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Interior rows of up to APPLY_BATCH_MAX blocks AND the perimeter rows of the rank in ONE launch (single-process handles: the
// perimeter rows read nothing the interior pass writes, and nothing has to arrive from another rank in between).  Small meshes --
// the reference's own inputs: 8 blocks, 25-38 k nodes -- are bound by dependent kernel launches (~5 us each), and the two passes
// are each a handful of memory latencies long: side by side they cost one launch and one of those chains instead of two.
template <int MODE, int DOT, bool HAS_PQ>
__global__ __launch_bounds__(256) void k_apply_edge_batch(ApplyBatch B, int total_interior, EdgeRowsDev e, const double2* __restrict__ in,
                                                          const double2* __restrict__ xk, const double2* __restrict__ pq,
                                                          const double2* __restrict__ aux, double2* __restrict__ out, double* edge_partials) {
    if (static_cast<int>(blockIdx.x) < total_interior) {   // workgroup-uniform
        int k = 0;
#pragma unroll
        for (int q = 1; q < APPLY_BATCH_MAX; ++q)
            if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
        apply_tile<MODE, DOT, false, HAS_PQ, 3, true>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], static_cast<int>(blockIdx.x) - B.start[k]);
        return;
    }
    const int wg = static_cast<int>(blockIdx.x) - total_interior;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    edge_rows_wg<MODE, DOT>(e, wg, threadIdx.x, in, xk, pq, aux, out, 0.0, acc);
    if (DOT != DOT_NONE) block_partials<256, dot_columns(DOT)>(acc, edge_partials + static_cast<size_t>(wg) * MAX_PARTIALS);   // waves 2, 3 add exact zeros
}

// An apply of a BiCGStab iteration with the vector update in front of it folded in (VirtualIn, tm_kernels.h):
//   VK_S  second apply: t = D^-1 A s, s = r - alpha v formed as the rows enter the window; partial sums t.s, t.t and ||s||^2
//         (DOT_IN_SS); s itself is never stored -- k_xr_update_vs forms it again.
//   VK_P  first apply: v' = D^-1 A p', p' = r + beta (p - omega v) formed the same way AND stored (V.pout: the next iteration and
//         k_xr_update_vs read it); partial sums r_hat . v' (DOT_AUX).  p' and v' are new arrays: neighbouring workgroups still
//         read the old p and v in their halos.
// The scalars come from the scalar block, advanced by the pending steps where there are any (LazyScalars).
// total_interior < 0: interior rows only (the perimeter rows follow in k_edge_rows_vk: large meshes).
template <int VK>
__device__ __forceinline__ VirtualArgs virtual_args(const VirtualIn& V, const KrylovScalars* S) {
    VirtualArgs A;
    A.in2 = V.in2;
    A.in3 = V.in3;
    A.in4 = V.in4;
    A.pout = V.pout;
    A.rout = V.rout;
    A.uio = V.uio;
    if (VK == VK_S || VK == VK_S2) {
        A.va = make_double2(S->alpha[0], S->alpha[1]);
    } else if (VK == VK_P) {
        A.va = make_double2(S->beta[0], S->beta[1]);
        A.vb = make_double2(S->omega[0], S->omega[1]);
    } else {
        A.va = make_double2(S->alpha[0], S->alpha[1]);
        A.vb = make_double2(S->omega[0], S->omega[1]);
        A.vc = make_double2(S->beta[0], S->beta[1]);
    }
    return A;
}
template <int VK>
struct VirtualDot {
    static constexpr int value = VK == VK_S ? DOT_IN_SS : (VK == VK_S2 ? DOT_B2 : DOT_AUX);
};
// MINW = workgroups per CU the register allocation must allow.  VK_R needs ~290 registers: with room for two workgroups per CU it
// spills 37-70 of them to scratch, which a small mesh (one round of workgroups, more of them than CUs) prefers to waiting for a
// second round (T106: 28.0 against 29.7 us per iteration); a large mesh prefers no spills (4096^2: 874 against 890 us), and so does
// a launch of at most one workgroup per CU (a 256^2 block: 20.7 against 22.1 us).
template <bool HAS_PQ, int VK, int MINW, bool OV = false>
__global__ __launch_bounds__(256, MINW) void k_apply_vk(ApplyBatch B, int total_interior, EdgeRowsDev e, VirtualIn V, const double2* __restrict__ xk,
                                                  const double2* __restrict__ pq, double2* __restrict__ out, double* edge_partials, LazyScalars L) {
    constexpr int DOT = VirtualDot<VK>::value;
    const KrylovScalars* S = lazy_scalars<256>(L);
    const VirtualArgs A = virtual_args<VK>(V, S);
    if (total_interior < 0 || static_cast<int>(blockIdx.x) < total_interior) {   // workgroup-uniform
        int k = 0;
#pragma unroll
        for (int q = 1; q < APPLY_BATCH_MAX; ++q)
            if (q < B.n && static_cast<int>(blockIdx.x) >= B.start[q]) k = q;
        apply_tile<MODE_SCALED, DOT, false, HAS_PQ, 3, true, VK, OV>(B.b[k], B.RI[k], B.nSG[k], B.nRC[k], static_cast<int>(blockIdx.x) - B.start[k], A.va, A.vb, A.vc);
        return;
    }
    const int wg = static_cast<int>(blockIdx.x) - total_interior;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    edge_rows_wg<MODE_SCALED, DOT, VK>(e, wg, threadIdx.x, V.in, xk, pq, V.aux, out, 0.0, acc, A);
    block_partials<256, (VK == VK_R ? 4 : dot_columns(DOT))>(acc, edge_partials + static_cast<size_t>(wg) * MAX_PARTIALS);
}
template <int VK>
__global__ __launch_bounds__(EDGE_BLOCK) void k_edge_rows_vk(EdgeRowsDev e, VirtualIn V, const double2* __restrict__ xk, const double2* __restrict__ pq,
                                                             double2* __restrict__ out, double* partials, const KrylovScalars* __restrict__ S) {
    constexpr int DOT = VirtualDot<VK>::value;
    const VirtualArgs A = virtual_args<VK>(V, S);
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    edge_rows_wg<MODE_SCALED, DOT, VK>(e, blockIdx.x, threadIdx.x, V.in, xk, pq, V.aux, out, 0.0, acc, A);
    block_partials<EDGE_BLOCK, (VK == VK_R ? 4 : dot_columns(DOT))>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}


hipError_t launch_edge_rows(const EdgeRowsDev& e, const double2* in, const double2* xk, const double2* pq, const double2* aux,
                            double2* out, double omega, int mode, int dot, double* partials, hipStream_t st, uint32_t* signal) {
    if (e.nrows == 0) return signal ? launch_queue_signal(signal, st) : hipSuccess;
    const dim3 grid(e.nwg), block(EDGE_BLOCK);
#define TM_EDGE(M, D)                                                                                                  \
    if (mode == M && dot == D) {                                                                                       \
        hipLaunchKernelGGL((k_edge_rows<M, D>), grid, block, 0, st, e, in, xk, pq, aux, out, omega, partials, signal); \
        return hipGetLastError();                                                                                      \
    }
    TM_EDGE(MODE_RAW, DOT_NONE)
    TM_EDGE(MODE_SCALED, DOT_NONE)
    TM_EDGE(MODE_SCALED, DOT_AUX)
    TM_EDGE(MODE_SCALED, DOT_AUX2)
    TM_EDGE(MODE_SCALED, DOT_IN)
    TM_EDGE(MODE_RESID, DOT_OUT2)
    TM_EDGE(MODE_RESID, DOT_NONE)
    TM_EDGE(MODE_RELAX, DOT_DELTA)
    TM_EDGE(MODE_RELAX, DOT_NONE)
#undef TM_EDGE
    return hipErrorInvalidValue;
}

// One launch for the interior rows of `n` <= APPLY_BATCH_MAX blocks (Krylov flavour: the operator acts on a vector other than the
// frozen field) and the perimeter rows `e`; hipErrorNotSupported when the combination has no merged kernel (caller falls back).
hipError_t launch_apply_edge_blocks(const ApplyBlock* blocks, int n, int mode, int dot, const EdgeRowsDev& e, const double2* in, const double2* xk,
                                    const double2* pq, const double2* aux, double2* out, double* edge_partials, hipStream_t st) {
    if (n > APPLY_BATCH_MAX || e.nrows == 0) return hipErrorNotSupported;
    ApplyBatch B;
    B.n = 0;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        const ApplyBlock& a = blocks[k];
        if (a.in == a.xk) return hipErrorNotSupported;   // field mode has its own kernels
        if (a.ni < 3 || a.nj < 3) continue;
        const int q = B.n++;
        B.b[q] = a;
        B.RI[q] = rows_per_chunk(a.ni, a.nj, a.rows);
        B.nSG[q] = (a.nj + 255) / 256;
        B.nRC[q] = (a.ni - 2 + B.RI[q] - 1) / B.RI[q];
        B.start[q] = total;
        total += B.nSG[q] * B.nRC[q];
    }
    for (int q = B.n; q < APPLY_BATCH_MAX; ++q) B.start[q] = total;
    const dim3 grid(total + e.nwg), block(256);
    const bool has_pq = pq != nullptr;
#define TM_AE(M, D)                                                                                                                       \
    if (mode == M && dot == D) {                                                                                                          \
        if (has_pq) hipLaunchKernelGGL((k_apply_edge_batch<M, D, true>), grid, block, 0, st, B, total, e, in, xk, pq, aux, out, edge_partials);   \
        else hipLaunchKernelGGL((k_apply_edge_batch<M, D, false>), grid, block, 0, st, B, total, e, in, xk, pq, aux, out, edge_partials);         \
        return hipGetLastError();                                                                                                         \
    }
    TM_AE(MODE_SCALED, DOT_AUX)
    TM_AE(MODE_SCALED, DOT_IN)
    TM_AE(MODE_RESID, DOT_OUT2)
#undef TM_AE
    return hipErrorNotSupported;
}

hipError_t launch_apply_virtual(const ApplyBlock* blocks, int n, const EdgeRowsDev& e, const VirtualIn& V, const double2* xk, const double2* pq, double2* out,
                                double* edge_partials, const LazyScalars& scal, hipStream_t st, bool overlap) {
    if (overlap && V.kind != VK_R && V.kind != VK_S2) return hipErrorInvalidValue;   // the two kernels of the two-kernel iteration
    if (e.nrows == 0 || V.kind < VK_S || V.kind > VK_S2) return hipErrorInvalidValue;
    const bool has_pq = pq != nullptr;
    const bool merged = n <= APPLY_BATCH_MAX;   // one launch for everything; else interior groups first, perimeter rows last
    LazyScalars L = scal;
    for (int first = 0; first < n; first += APPLY_BATCH_MAX) {
        ApplyBatch B;
        B.n = 0;
        int total = 0;
        for (int k = first; k < n && B.n < APPLY_BATCH_MAX; ++k) {
            const ApplyBlock& a = blocks[k];
            if (a.ni < 3 || a.nj < 3) continue;
            const int q = B.n++;
            B.b[q] = a;
            B.RI[q] = rows_per_chunk(a.ni, a.nj, a.rows);
            B.nSG[q] = overlap ? overlap_strip_groups(a.nj) : (a.nj + 255) / 256;
            B.nRC[q] = (a.ni - 2 + B.RI[q] - 1) / B.RI[q];
            B.start[q] = total;
            total += B.nSG[q] * B.nRC[q];
        }
        for (int q = B.n; q < APPLY_BATCH_MAX; ++q) B.start[q] = total;
        const dim3 grid(merged ? total + e.nwg : total), block(256);
        if (grid.x == 0) continue;
        const int ti = merged ? total : -1;
        if (overlap) {   // 62-column strips: no halo registers, two workgroups per CU without spills
            if (V.kind == VK_R) {
                if (has_pq) hipLaunchKernelGGL((k_apply_vk<true, VK_R, 2, true>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L);
                else hipLaunchKernelGGL((k_apply_vk<false, VK_R, 2, true>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L);
            } else {
                if (has_pq) hipLaunchKernelGGL((k_apply_vk<true, VK_S2, 2, true>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L);
                else hipLaunchKernelGGL((k_apply_vk<false, VK_S2, 2, true>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L);
            }
        } else
#define TM_VK(PQ, K) hipLaunchKernelGGL((k_apply_vk<PQ, K, 2>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L)
#define TM_VK1(PQ, K) hipLaunchKernelGGL((k_apply_vk<PQ, K, 1>), grid, block, 0, st, B, ti, e, V, xk, pq, out, edge_partials, L)
        if (V.kind == VK_S) {
            if (has_pq) TM_VK(true, VK_S);
            else TM_VK(false, VK_S);
        } else if (V.kind == VK_P) {
            if (has_pq) TM_VK(true, VK_P);
            else TM_VK(false, VK_P);
        } else if (V.kind == VK_R) {
            if (grid.x > 2048 || grid.x <= 256) {   // large meshes; and launches of at most one workgroup per CU (nothing to fit beside it)
                if (has_pq) TM_VK1(true, VK_R);
                else TM_VK1(false, VK_R);
            } else {
                if (has_pq) TM_VK(true, VK_R);
                else TM_VK(false, VK_R);
            }
        } else {
            if (has_pq) TM_VK(true, VK_S2);
            else TM_VK(false, VK_S2);
        }
#undef TM_VK
#undef TM_VK1
        const hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        if (L.nsteps) {   // the first launch has applied and published the pending steps: later ones read the published block
            L.S_in = L.S_out;
            L.S_out = nullptr;
            L.nsteps = 0;
        }
    }
    if (!merged) {
        if (V.kind == VK_S) hipLaunchKernelGGL(k_edge_rows_vk<VK_S>, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, V, xk, pq, out, edge_partials, L.S_in);
        else if (V.kind == VK_P) hipLaunchKernelGGL(k_edge_rows_vk<VK_P>, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, V, xk, pq, out, edge_partials, L.S_in);
        else if (V.kind == VK_R) hipLaunchKernelGGL(k_edge_rows_vk<VK_R>, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, V, xk, pq, out, edge_partials, L.S_in);
        else hipLaunchKernelGGL(k_edge_rows_vk<VK_S2>, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, V, xk, pq, out, edge_partials, L.S_in);
        return hipGetLastError();
    }
    return hipSuccess;
}


// ------------------------------------------------------------------------------------------
// Multigrid transfer kernels (N4): one thread per node of the level written, j fastest.  All HBM-bound and small next to
// the smoothing sweeps (K2 in MODE_MG_*): the fine level dominates, each coarser level has 1/4 of the nodes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int mg_fine_of(int c, int coarsened, int nf) { return coarsened ? min(2 * c, nf - 1) : c; }

__global__ __launch_bounds__(256) void k_mg_inject(const double2* __restrict__ fine, double2* __restrict__ coarse, MgPair g, double sx, double sy) {
    const int cj = blockIdx.x * 256 + threadIdx.x;
    if (cj >= g.njc) return;
    for (int ci = blockIdx.y; ci < g.nic; ci += gridDim.y) {   // gridDim.y is capped at 65535
        const double2 v = fine[static_cast<size_t>(mg_fine_of(ci, g.ci, g.nif)) * g.njf + mg_fine_of(cj, g.cj, g.njf)];
        coarse[static_cast<size_t>(ci) * g.njc + cj] = make_double2(sx * v.x, sy * v.y);
    }
}
hipError_t launch_mg_inject(const double2* fine, double2* coarse, const MgPair& g, double sx, double sy, hipStream_t st) {
    hipLaunchKernelGGL(k_mg_inject, dim3((g.njc + 255) / 256, std::min(g.nic, 65535)), dim3(256), 0, st, fine, coarse, g, sx, sy);
    return hipGetLastError();
}

// The levels' equations are row-equilibrated (D^-1 A e = f), and D does not scale uniformly between levels, so the residual
// travels unscaled: with index spacings s_i, s_j in {1,2} the rediscretised operator satisfies A_c = (s_i s_j)^2 A_f in the
// smooth limit (every term: g22 d_xixi, g11 d_etaeta, g12 d_xieta), hence f_c = (s_i s_j)^2 R(a_ii^f rho_f) / a_ii^c.
__global__ __launch_bounds__(256) void k_mg_restrict(const double2* __restrict__ rf, const double2* __restrict__ Xc, double2* __restrict__ fc, MgPair g) {
    const int cj = blockIdx.x * 256 + threadIdx.x + 1;   // interior coarse nodes
    if (cj > g.njc - 2) return;
    for (int ci = blockIdx.y + 1; ci <= g.nic - 2; ci += gridDim.y) {
    const int fi = g.ci ? 2 * ci : ci, fj = g.cj ? 2 * cj : cj;   // interior coarse nodes never hit the short last cell's clamp
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
        if (!g.ci && di != 0) continue;
        const double wi = g.ci ? (di == 0 ? 0.5 : 0.25) : 1.0;
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            if (!g.cj && dj != 0) continue;
            const double w = wi * (g.cj ? (dj == 0 ? 0.5 : 0.25) : 1.0);
            const double2 v = rf[static_cast<size_t>(fi + di) * g.njf + (fj + dj)];
            acc.x = fma(w, v.x, acc.x);
            acc.y = fma(w, v.y, acc.y);
        }
    }
    const size_t o = static_cast<size_t>(ci) * g.njc + cj;
    const double2 xp = Xc[o + g.njc], xm = Xc[o - g.njc], xr = Xc[o + 1], xl = Xc[o - 1];
    const double dxx = xp.x - xm.x, dxy = xp.y - xm.y, dex = xr.x - xl.x, dey = xr.y - xl.y;
    const double aii = -0.5 * (fma(dxx, dxx, dxy * dxy) + fma(dex, dex, dey * dey));
    const double k = static_cast<double>((g.ci ? 4 : 1) * (g.cj ? 4 : 1)) / ((aii == 0.0) ? 1.0 : aii);
    fc[o] = make_double2(k * acc.x, k * acc.y);
    }
}
hipError_t launch_mg_restrict(const double2* rf, const double2* Xc, double2* fc, const MgPair& g, hipStream_t st) {
    if (g.nic < 3 || g.njc < 3) return hipSuccess;
    hipLaunchKernelGGL(k_mg_restrict, dim3((g.njc - 2 + 255) / 256, std::min(g.nic - 2, 65535)), dim3(256), 0, st, rf, Xc, fc, g);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_mg_prolong_add(const double2* __restrict__ ec, double2* __restrict__ ef, MgPair g) {
    const int j = blockIdx.x * 256 + threadIdx.x + 1;   // interior fine nodes
    if (j > g.njf - 2) return;
    for (int i = blockIdx.y + 1; i <= g.nif - 2; i += gridDim.y) {
    // per direction: an even fine index (or an uncoarsened direction) coincides with a coarse node, an odd one lies midway
    int ci0 = i, ci1 = i, cj0 = j, cj1 = j;
    if (g.ci) {
        ci0 = i >> 1;
        ci1 = (i + 1) >> 1;
    }
    if (g.cj) {
        cj0 = j >> 1;
        cj1 = (j + 1) >> 1;
    }
    const double2 a = ec[static_cast<size_t>(ci0) * g.njc + cj0], b = ec[static_cast<size_t>(ci0) * g.njc + cj1];
    const double2 c = ec[static_cast<size_t>(ci1) * g.njc + cj0], d = ec[static_cast<size_t>(ci1) * g.njc + cj1];
    double2 e = ef[static_cast<size_t>(i) * g.njf + j];
    e.x += 0.25 * ((a.x + b.x) + (c.x + d.x));   // coinciding indices just repeat a value: weights 1, 1/2 1/2 or 1/4 x 4
    e.y += 0.25 * ((a.y + b.y) + (c.y + d.y));
    ef[static_cast<size_t>(i) * g.njf + j] = e;
    }
}
hipError_t launch_mg_prolong_add(const double2* ec, double2* ef, const MgPair& g, hipStream_t st) {
    if (g.nif < 3 || g.njf < 3) return hipSuccess;
    hipLaunchKernelGGL(k_mg_prolong_add, dim3((g.njf - 2 + 255) / 256, std::min(g.nif - 2, 65535)), dim3(256), 0, st, ec, ef, g);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_mg_scale(const double2* __restrict__ f, double2* __restrict__ out, int ni, int nj, double omega) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nj) return;
    for (int i = blockIdx.y; i < ni; i += gridDim.y) {
        const bool interior = i >= 1 && i <= ni - 2 && j >= 1 && j <= nj - 2;
        const size_t o = static_cast<size_t>(i) * nj + j;
        const double2 v = interior ? f[o] : make_double2(0.0, 0.0);
        out[o] = make_double2(omega * v.x, omega * v.y);
    }
}
hipError_t launch_mg_scale(const double2* f, double2* out, int ni, int nj, double omega, hipStream_t st) {
    hipLaunchKernelGGL(k_mg_scale, dim3((nj + 255) / 256, std::min(ni, 65535)), dim3(256), 0, st, f, out, ni, nj, omega);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K8  soa_planes: the export transpose of cgns.zig:75-104 (and :106-154 for P,Q) -- node (i,j) of the
//     interleaved block (index i*nj + j) goes to plane element j*ni + i (i fastest), x and y (or P and Q)
//     into separate planes.  32 x 32 tiles through LDS: reads coalesced along j, writes coalesced along i.
//     HBM-bound, 32 B/node (16 read + 16 written).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_soa_planes(const double2* __restrict__ in, double* __restrict__ plane0, double* __restrict__ plane1, int ni,
                                                    int nj) {
    __shared__ double2 tile[32][33];   // +1: the transposed read walks a column
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int i = i0 + ty + r, j = j0 + tx;
        if (i < ni && j < nj) tile[ty + r][tx] = in[static_cast<size_t>(i) * nj + j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int j = j0 + ty + r, i = i0 + tx;
        if (i < ni && j < nj) {
            const double2 v = tile[tx][ty + r];
            const size_t o = static_cast<size_t>(j) * ni + i;
            plane0[o] = v.x;
            plane1[o] = v.y;
        }
    }
}
hipError_t launch_soa_planes(const double2* in, double* plane0, double* plane1, int ni, int nj, hipStream_t st) {
    const dim3 grid((nj + 31) / 32, (ni + 31) / 32), block(256);
    hipLaunchKernelGGL(k_soa_planes, grid, block, 0, st, in, plane0, plane1, ni, nj);
    return hipGetLastError();
}

// right-hand side of the perimeter rows (interior rows have b = 0)
__global__ __launch_bounds__(EDGE_BLOCK) void k_edge_rhs(EdgeRowsDev e, const double2* __restrict__ xk, const double2* __restrict__ pq,
                                                         double2* __restrict__ rhs_out, int scaled, double* partials) {
    const EdgeRun& R = e.runs[__builtin_amdgcn_readfirstlane(e.wg_run[blockIdx.x])];
    const int k = __builtin_amdgcn_readfirstlane(e.wg_k0[blockIdx.x]) + static_cast<int>(threadIdx.x);
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (k < R.count) {
        const int row = R.row0 + k * R.row_stride;
        double rhs_x, rhs_y, diag_x, diag_y;
        if (R.kind == 1) {
            const double2 im1_j = xk[R.met0[0] + k * R.met_stride[0]], ip1_j = xk[R.met0[1] + k * R.met_stride[1]];
            const double2 i_jm1 = xk[R.met0[2] + k * R.met_stride[2]];
            double2 i_jp1 = xk[R.met0[3] + k * R.met_stride[3]];
            const bool periodic = R.flags & 1;
            const double per_x = R.per[0], per_y = R.per[1];
            if (periodic) {
                i_jp1.x = i_jp1.x + (-per_x);
                i_jp1.y = i_jp1.y + (-per_y);
            }
            const double2 cf = pq ? pq[row] : make_double2(0.0, 0.0);
            const double P = periodic ? cf.x : cf.y, Q = periodic ? cf.y : cf.x;
            double c[9];
            stencil_coefs<true>(im1_j, ip1_j, i_jm1, i_jp1, P, Q, c);
            diag_x = diag_y = c[S_I_J];
            const double cs = c[S_IM1_JP1] + c[S_I_JP1] + c[S_IP1_JP1];
            rhs_x = periodic ? per_x * cs : 0.0;
            rhs_y = periodic ? per_y * cs : 0.0;
        } else {
            const int self = R.self;
            diag_x = 0.0;
            diag_y = 0.0;
#pragma unroll
            for (int q = 0; q < 9; ++q)
                if (q == self) {
                    diag_x = R.cx[q];
                    diag_y = R.cy[q];
                }
            rhs_x = e.rhs[2 * (R.first + k)];
            rhs_y = e.rhs[2 * (R.first + k) + 1];
        }
        const double bx = rhs_x * ((diag_x == 0.0) ? 1.0 : 1.0 / diag_x);
        const double by = rhs_y * ((diag_y == 0.0) ? 1.0 : 1.0 / diag_y);
        if (rhs_out) rhs_out[row] = scaled ? make_double2(bx, by) : make_double2(rhs_x, rhs_y);
        acc[0] = bx * bx;
        acc[1] = by * by;
    }
    if (partials) block_partials<EDGE_BLOCK, 2>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}
hipError_t launch_edge_rhs(const EdgeRowsDev& e, const double2* xk, const double2* pq, double2* rhs_out, int scaled, double* partials,
                           hipStream_t st) {
    if (e.nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_edge_rhs, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, xk, pq, rhs_out, scaled, partials);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// queue-to-queue ordering through device memory (see Smoother::relax_pairs_pipelined)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_queue_signal(uint32_t* counter) {
    // the kernels before this one in the queue have completed (in-order queue, end-of-kernel release); publish at agent scope
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(64) void k_queue_wait(const uint32_t* counter, uint32_t target, uint32_t* error, long long limit_ticks) {
    // relaxed polls (an acquire per poll would invalidate this XCD's L2 every microsecond under the interior pass); the kernels
    // behind this one start with the usual start-of-kernel acquire and see what the signalling queue had completed
    if (threadIdx.x == 0) spin_until(counter, target, error, limit_ticks);
}
// both in one launch (a kernel boundary less on the handle's stream): announce what precedes, then wait for the other queue
__global__ __launch_bounds__(64) void k_queue_signal_wait(uint32_t* counter, const uint32_t* other, uint32_t target, uint32_t* error, long long limit_ticks) {
    if (threadIdx.x != 0) return;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(other, target, error, limit_ticks);
}
hipError_t launch_queue_signal_wait(uint32_t* counter, const uint32_t* other, uint32_t target, uint32_t* error, hipStream_t st, long long limit_ticks) {
    hipLaunchKernelGGL(k_queue_signal_wait, dim3(1), dim3(64), 0, st, counter, other, target, error, limit_ticks);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------
// The assembled system (introspection: tm_smoother_assemble_csr, tm_smoother_apply_reference_order)
// ------------------------------------------------------------------------------------------
template <bool HAS_PQ>
__global__ __launch_bounds__(256) void k_assemble_interior(const double2* __restrict__ xk, const double2* __restrict__ pq, int ni, int nj,
                                                           const int32_t* __restrict__ row_ptr, double* __restrict__ vx, double* __restrict__ vy) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < 1 || j > nj - 2) return;
    for (int i = 1 + static_cast<int>(blockIdx.y); i <= ni - 2; i += gridDim.y) {
        const size_t at = static_cast<size_t>(i) * nj + j;
        double c[9];
        const double2 cf = HAS_PQ ? pq[at] : make_double2(0.0, 0.0);
        stencil_coefs<HAS_PQ>(xk[at - nj], xk[at + nj], xk[at - 1], xk[at + 1], cf.x, cf.y, c);   // im1_j, ip1_j, i_jm1, i_jp1; (P, Q): smooth.zig:936-945
        const int e = row_ptr[at];
        // ascending column id = (i-1, j-1..j+1), (i, j-1..j+1), (i+1, j-1..j+1): smooth.zig:946-955
        const double v[9] = {c[S_IM1_JM1], c[S_IM1_J], c[S_IM1_JP1], c[S_I_JM1], c[S_I_J], c[S_I_JP1], c[S_IP1_JM1], c[S_IP1_J], c[S_IP1_JP1]};
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            vx[e + q] = v[q];
            vy[e + q] = v[q];
        }
    }
}
hipError_t launch_assemble_interior(const double2* xk, const double2* pq, int ni, int nj, const int32_t* row_ptr, double* vx, double* vy, hipStream_t st) {
    if (ni < 3 || nj < 3) return hipSuccess;
    const dim3 grid((nj + 255) / 256, std::min(ni - 2, 4096));
    if (pq) hipLaunchKernelGGL(k_assemble_interior<true>, grid, dim3(256), 0, st, xk, pq, ni, nj, row_ptr, vx, vy);
    else hipLaunchKernelGGL(k_assemble_interior<false>, grid, dim3(256), 0, st, xk, pq, ni, nj, row_ptr, vx, vy);
    return hipGetLastError();
}
__global__ __launch_bounds__(EDGE_BLOCK) void k_assemble_edge(EdgeRowsDev e, const double2* __restrict__ xk, const double2* __restrict__ pq,
                                                              const int32_t* __restrict__ row_ptr, double* __restrict__ vx, double* __restrict__ vy) {
    const EdgeRun& R = e.runs[__builtin_amdgcn_readfirstlane(e.wg_run[blockIdx.x])];
    const int k = __builtin_amdgcn_readfirstlane(e.wg_k0[blockIdx.x]) + static_cast<int>(threadIdx.x);
    if (k >= R.count) return;
    const int row = R.row0 + k * R.row_stride;
    const int e0 = row_ptr[row];
    if (R.kind == 1 /* smoothed */) {   // as edge_row_eval: smooth.zig:1029-1084
        const double2 im1_j = xk[R.met0[0] + k * R.met_stride[0]], ip1_j = xk[R.met0[1] + k * R.met_stride[1]];
        const double2 i_jm1 = xk[R.met0[2] + k * R.met_stride[2]];
        double2 i_jp1 = xk[R.met0[3] + k * R.met_stride[3]];
        const bool periodic = R.flags & 1;
        if (periodic) {
            i_jp1.x = i_jp1.x + (-R.per[0]);
            i_jp1.y = i_jp1.y + (-R.per[1]);
        }
        const double2 cf = pq ? pq[row] : make_double2(0.0, 0.0);
        const double P = periodic ? cf.x : cf.y, Q = periodic ? cf.y : cf.x;
        double c[9];
        stencil_coefs<true>(im1_j, ip1_j, i_jm1, i_jp1, P, Q, c);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double ck = pick9(c, R.slot[q]);
            vx[e0 + q] = ck;
            vy[e0 + q] = ck;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
        if (q < R.ncols) {
            vx[e0 + q] = R.cx[q];
            vy[e0 + q] = R.cy[q];
        }
}
hipError_t launch_assemble_edge(const EdgeRowsDev& e, const double2* xk, const double2* pq, const int32_t* row_ptr, double* vx, double* vy, hipStream_t st) {
    if (e.nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_assemble_edge, dim3(e.nwg), dim3(EDGE_BLOCK), 0, st, e, xk, pq, row_ptr, vx, vy);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void k_csr_product(int64_t n, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col, const double* __restrict__ vx,
                                                     const double* __restrict__ vy, const double2* __restrict__ in, double2* __restrict__ out) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n) return;
    double sx = 0.0, sy = 0.0;
    for (int k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {   // sum += lhs_values[k] * x[lhs_i[k]], BiCGStab.zig:424-435
        const double2 w = in[col[k]];
        sx += vx[k] * w.x;
        sy += vy[k] * w.y;
    }
    out[row] = make_double2(sx, sy);
}
hipError_t launch_csr_product(int64_t n, const int32_t* row_ptr, const int32_t* col, const double* vx, const double* vy, const double2* in, double2* out, hipStream_t st) {
    hipLaunchKernelGGL(k_csr_product, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, n, row_ptr, col, vx, vy, in, out);
    return hipGetLastError();
}

// measurement support (libtm_hip_dbg.so's null transport): one wave that keeps its queue busy for `us` microseconds of the constant
// 100 MHz clock -- the device time of a halo exchange that moves nothing, so that a rank's schedule can be timed with the exchange
// ON its chain without peers
__global__ __launch_bounds__(64) void k_delay(long long ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_delay_us(double us, hipStream_t st) {
    if (!(us > 0.0)) return hipSuccess;
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, static_cast<long long>(us * 100.0));
    return hipGetLastError();
}
hipError_t launch_queue_signal(uint32_t* counter, hipStream_t st) {
    hipLaunchKernelGGL(k_queue_signal, dim3(1), dim3(64), 0, st, counter);
    return hipGetLastError();
}
hipError_t launch_queue_wait(const uint32_t* counter, uint32_t target, uint32_t* error, hipStream_t st, long long limit_ticks) {
    hipLaunchKernelGGL(k_queue_wait, dim3(1), dim3(64), 0, st, counter, target, error, limit_ticks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
// Sums of `nwg` partial rows by a 256-thread workgroup -- ONE association whatever the caller: the launched finalize kernels and
// every workgroup of a kernel that applies pending scalar steps must agree to the bit.  Thread t adds rows t, t + 256, ... (all
// columns of a row: the loads of a trip are independent, and a small mesh needs one or two trips); thread (g, k) = (t / 8, t % 8)
// adds column k of the eight thread sums 8 g .. 8 g + 7; thread k adds the 32 group sums of its column in ascending g.  Three
// barriers and 8 + 32 dependent additions instead of an eight-level tree with a barrier per level: the lazy prologue sits on the
// critical path of every kernel of a small mesh.  Result in out[0..MAX_PARTIALS) (LDS), visible to all threads on return.
struct PartialSumsLds {
    double rows[256][MAX_PARTIALS];
    double groups[32][MAX_PARTIALS];
};
static_assert(MAX_PARTIALS == 8, "sum_partial_rows lays 256 threads out as 32 groups x 8 columns");
__device__ __forceinline__ void sum_partial_rows(const double* __restrict__ partials, int nwg, PartialSumsLds& sh, double* out) {
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nwg; i += 256) {
#pragma unroll
        for (int k = 0; k < MAX_PARTIALS; ++k) acc[k] += partials[static_cast<size_t>(i) * MAX_PARTIALS + k];
    }
#pragma unroll
    for (int k = 0; k < MAX_PARTIALS; ++k) sh.rows[threadIdx.x][k] = acc[k];
    __syncthreads();
    {
        const int g = threadIdx.x >> 3, k = threadIdx.x & 7;
        double a = sh.rows[8 * g][k];
#pragma unroll
        for (int q = 1; q < 8; ++q) a += sh.rows[8 * g + q][k];
        sh.groups[g][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < MAX_PARTIALS) {
        double t = sh.groups[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 32; ++q) t += sh.groups[q][threadIdx.x];
        out[threadIdx.x] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partials, int nwg, double* __restrict__ red) {
    __shared__ PartialSumsLds sh;
    __shared__ double tot[MAX_PARTIALS];
    sum_partial_rows(partials, nwg, sh, tot);
    if (threadIdx.x < MAX_PARTIALS) red[threadIdx.x] = tot[threadIdx.x];
}
hipError_t launch_finalize(const double* partials, int nwg, double* red, hipStream_t st) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, partials, nwg, red);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K3 fused BiCGStab vector kernels (recurrences of BiCGStab.zig:279-370 on D^-1 A, no
// preconditioner vector: the Jacobi scaling lives in K2).  Scalars stay on the device.
// ------------------------------------------------------------------------------------------
int vec_nwg(int64_t n) {
    int64_t g = (n + VEC_BLOCK - 1) / VEC_BLOCK;
    const int64_t cap = 256 * 8;   // 8 workgroups per CU, grid-stride beyond
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

__device__ __forceinline__ void scalar_update_body(KrylovScalars* S, const double* red, int step, double rtol, double atol, int c) {
    const double tiny = 1e-290;
    if (step == STEP_TOL) {
        if (rtol < 0.0) {   // TM_OPT_RTOL_INITIAL: the tolerance is relative to the solve's INITIAL residual, known at STEP_INIT -- parked until then
            S->tol2[c] = -(rtol * rtol);
            S->rr[c] = atol * atol;
            return;
        }
        const double tol = fmax(atol, rtol * sqrt(red[c]));
        S->tol2[c] = tol * tol;
        return;
    }
    if ((step == STEP_INIT || step == STEP_INIT2) && S->tol2[c] < 0.0) S->tol2[c] = fmax(S->rr[c], -S->tol2[c] * red[c]);   // (a restart keeps it)
    if (step == STEP_INIT) {   // red[0..1] = ||r0||^2 ; r_hat = r0 so rho = ||r0||^2
        const double rr = red[c];
        S->rr[c] = rr;
        S->rr0[c] = rr;
        S->rho[c] = rr;
        S->rho_old[c] = 1.0;
        S->alpha[c] = 1.0;
        S->omega[c] = 1.0;
        S->beta[c] = 0.0;   // p <- r (k_p_update selects on it: p and v need not be cleared; the fused forms multiply cleared arrays)
        S->early[c] = 0;
        const bool done = !(rr > S->tol2[c]);
        S->done[c] = done ? 1 : 0;
        if (done) S->alpha[c] = S->omega[c] = S->beta[c] = 0.0;
        if (c == 0) S->iters = 0;
        return;
    }
    if (step == STEP_INIT2) {   // STEP_INIT for the two-kernel iteration: nothing pending, the first p' is r
        const double rr = red[c];
        S->rr[c] = rr;
        S->rr0[c] = rr;
        S->rho[c] = rr;
        S->rho_old[c] = 1.0;
        S->alpha[c] = S->omega[c] = S->beta[c] = 0.0;
        S->early[c] = 0;
        S->done[c] = !(rr > S->tol2[c]) ? 1 : 0;
        if (c == 0) S->iters = 0;
        return;
    }
    if (S->done[c]) return;
    if (step == STEP_A2) {   // red[0..1] = r_hat . v', red[2..3] = ||r'||^2 of the update that rode in front of this apply
        const double sigma = red[c], rr = red[2 + c];
        S->rr[c] = rr;
        int done = 0;
        if (rr <= S->tol2[c]) done = 1;
        else if (S->early[c] == 2 || !(fabs(sigma) > tiny)) done = 2;
        S->early[c] = 0;
        if (done) {   // the x / r of this component are final: every later update is a no-op
            S->done[c] = done;
            S->alpha[c] = S->omega[c] = S->beta[c] = 0.0;
        } else {
            S->alpha[c] = S->rho[c] / sigma;
        }
        if (c == 0) S->iters += 1;
        return;
    }
    if (step == STEP_B2) {   // red[0..1] = t.s, red[2..3] = t.t, red[4..5] = r_hat.s, red[6..7] = r_hat.t
        const double ts = red[c], tt = red[2 + c], hs = red[4 + c], ht = red[6 + c];
        double omega = 0.0;
        if (tt > tiny) omega = ts / tt;   // else: s is (numerically) zero -- r' = s, and STEP_A2 sees ||r'||^2 <= tol, or a breakdown
        const double rho_new = hs - omega * ht;   // = r_hat . (s - omega t) = r_hat . r'
        S->omega[c] = omega;
        S->rho_old[c] = S->rho[c];
        S->rho[c] = rho_new;
        if (!(fabs(omega) > tiny) || !(fabs(rho_new) > tiny)) {
            S->early[c] = 2;   // breakdown, unless the pending update turns out to have converged (STEP_A2)
            S->beta[c] = 0.0;
        } else {
            S->beta[c] = (rho_new / S->rho_old[c]) * (S->alpha[c] / omega);
        }
        return;
    }
    if (step == STEP_SIGMA) {   // red = r_hat . v
        const double sigma = red[c];
        if (!(fabs(sigma) > tiny)) {
            S->done[c] = 2;
            S->alpha[c] = S->omega[c] = S->beta[c] = 0.0;
        } else {
            S->alpha[c] = S->rho[c] / sigma;
        }
    } else if (step == STEP_SS) {   // red = ||s||^2
        S->rr[c] = red[c];
        S->early[c] = (red[c] <= S->tol2[c]) ? 1 : 0;
    } else if (step == STEP_SS_TSTT) {   // red[4..5] = ||s||^2, red[0..1] = t.s, red[2..3] = t.t: STEP_SS, then STEP_TSTT
        S->rr[c] = red[4 + c];
        S->early[c] = (red[4 + c] <= S->tol2[c]) ? 1 : 0;
        const double ts = red[c], tt = red[2 + c];
        if (S->early[c]) S->omega[c] = 0.0;
        else if (!(tt > tiny)) {
            S->omega[c] = 0.0;
            S->early[c] = 2;   // breakdown after this update
        } else S->omega[c] = ts / tt;
    } else if (step == STEP_TSTT) {   // red[0..1] = t.s, red[2..3] = t.t
        const double ts = red[c], tt = red[2 + c];
        if (S->early[c]) S->omega[c] = 0.0;
        else if (!(tt > tiny)) {
            S->omega[c] = 0.0;
            S->early[c] = 2;   // breakdown after this update
        } else S->omega[c] = ts / tt;
    } else if (step == STEP_RHO) {   // red[0..1] = r_hat . r, red[2..3] = ||r||^2
        const double rho_new = red[c], rr = red[2 + c];
        S->rho_old[c] = S->rho[c];
        S->rho[c] = rho_new;
        S->rr[c] = rr;
        int done = 0;
        if (S->early[c] == 1 || rr <= S->tol2[c]) done = 1;
        else if (S->early[c] == 2 || !(fabs(rho_new) > tiny) || !(fabs(S->omega[c]) > tiny)) done = 2;
        S->early[c] = 0;
        if (done) {
            S->done[c] = done;
            S->alpha[c] = S->omega[c] = S->beta[c] = 0.0;
        } else {
            S->beta[c] = (rho_new / S->rho_old[c]) * (S->alpha[c] / S->omega[c]);
        }
        if (c == 0) S->iters += 1;
    }
}
__global__ void k_scalar_update(KrylovScalars* S, const double* __restrict__ red, int step, double rtol, double atol) {
    if (threadIdx.x < 2) scalar_update_body(S, red, step, rtol, atol, threadIdx.x);
}
// finalize + scalar update in one launch (single-process handles: nothing has to be all-reduced in between)
__global__ __launch_bounds__(256) void k_finalize_scalar(const double* __restrict__ partials, int nwg, double* __restrict__ red, KrylovScalars* S, int step,
                                                         double rtol, double atol) {
    __shared__ PartialSumsLds sh;
    __shared__ double tot[MAX_PARTIALS];
    sum_partial_rows(partials, nwg, sh, tot);
    if (threadIdx.x < MAX_PARTIALS) red[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x < 2) scalar_update_body(S, tot, step, rtol, atol, threadIdx.x);
}
hipError_t launch_finalize_scalar(const double* partials, int nwg, double* red, KrylovScalars* S, int step, hipStream_t st, double rtol, double atol) {
    hipLaunchKernelGGL(k_finalize_scalar, dim3(1), dim3(256), 0, st, partials, nwg, red, S, step, rtol, atol);
    return hipGetLastError();
}
hipError_t launch_scalar_update(KrylovScalars* S, const double* red, int step, hipStream_t st, double rtol, double atol) {
    hipLaunchKernelGGL(k_scalar_update, dim3(1), dim3(64), 0, st, S, red, step, rtol, atol);
    return hipGetLastError();
}

// LazyScalars (tm_kernels.h): the scalars a vector kernel works with -- S_in as it stands, or S_in advanced by the pending steps.
template <int NT>
__device__ __forceinline__ const KrylovScalars* lazy_scalars(const LazyScalars& L) {
    if (L.nsteps == 0) return L.S_in;   // uniform
    __shared__ KrylovScalars shS;
    __shared__ PartialSumsLds shR;
    __shared__ double tot[MAX_PARTIALS];
    static_assert(NT == 256, "sum_partial_rows is written for 256 threads");
    constexpr int NWORDS = sizeof(KrylovScalars) / 4;
    static_assert(sizeof(KrylovScalars) % 4 == 0 && NWORDS <= NT, "scalar block is copied one word per thread");
    if (threadIdx.x < NWORDS) reinterpret_cast<uint32_t*>(&shS)[threadIdx.x] = reinterpret_cast<const uint32_t*>(L.S_in)[threadIdx.x];
    for (int q = 0; q < L.nsteps; ++q) {
        sum_partial_rows(L.st[q].partials, L.st[q].nwg, shR, tot);   // the sums of k_finalize: the same in every workgroup
        if (threadIdx.x < 2) scalar_update_body(&shS, tot, L.st[q].step, 0.0, 0.0, threadIdx.x);
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x < NWORDS) reinterpret_cast<uint32_t*>(L.S_out)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&shS)[threadIdx.x];
    return &shS;
}

// The vector kernels stream 3-7 arrays far larger than the caches (256 MiB each at 4096^2) and re-use nothing: every thread
// takes VEC_UNROLL elements per trip with all their loads issued before the first use (more requests in flight per wave), loads
// and stores are non-temporal.
constexpr int VEC_UNROLL = 2;
__device__ __forceinline__ double2 load_nt(const double2* src) {
    const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(src));
    return make_double2(v.x, v.y);
}

__global__ __launch_bounds__(VEC_BLOCK) void k_p_update(LazyScalars L, const double2* __restrict__ r,
                                                        double2* __restrict__ p, const double2* __restrict__ v, int64_t n) {
    const KrylovScalars* S = lazy_scalars<VEC_BLOCK>(L);
    const double bx = S->beta[0], by = S->beta[1], ox = S->omega[0], oy = S->omega[1];
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * VEC_UNROLL) {
        double2 ri[VEC_UNROLL], pi[VEC_UNROLL], vi[VEC_UNROLL];
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            ri[q] = load_nt(r + i);
            pi[q] = load_nt(p + i);
            vi[q] = load_nt(v + i);
        }
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            // BiCGStab.zig:310-312; beta = 0 (the first iteration, a finished component): p = r whatever p and v hold
            if (i < n) store_nt(p + i, make_double2(bx == 0.0 ? ri[q].x : ri[q].x + bx * (pi[q].x - ox * vi[q].x), by == 0.0 ? ri[q].y : ri[q].y + by * (pi[q].y - oy * vi[q].y)));
        }
    }
}
hipError_t launch_p_update(const LazyScalars& S, const double2* r, double2* p, const double2* v, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_p_update, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, S, r, p, v, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(VEC_BLOCK) void k_s_update(LazyScalars L, const double2* __restrict__ r,
                                                        const double2* __restrict__ v, double2* __restrict__ s, int64_t n,
                                                        double* partials) {
    const KrylovScalars* S = lazy_scalars<VEC_BLOCK>(L);
    const double ax = S->alpha[0], ay = S->alpha[1];
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * VEC_UNROLL) {
        double2 ri[VEC_UNROLL], vi[VEC_UNROLL];
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            ri[q] = load_nt(r + i);
            vi[q] = load_nt(v + i);
        }
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n) {
                const double2 si = make_double2(ri[q].x - ax * vi[q].x, ri[q].y - ay * vi[q].y);   // BiCGStab.zig:325-327
                store_nt(s + i, si);
                acc[0] += si.x * si.x;
                acc[1] += si.y * si.y;
            }
        }
    }
    block_partials<VEC_BLOCK, 2>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}
hipError_t launch_s_update(const LazyScalars& S, const double2* r, const double2* v, double2* s, int64_t n, double* partials,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_s_update, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, S, r, v, s, n, partials);
    return hipGetLastError();
}

__global__ __launch_bounds__(VEC_BLOCK) void k_xr_update(LazyScalars L, double2* __restrict__ u, const double2* p_hat,
                                                         const double2* s_hat, const double2* s, const double2* __restrict__ t, double2* r,
                                                         const double2* __restrict__ r_hat, int64_t n, double* partials) {
    const KrylovScalars* S = lazy_scalars<VEC_BLOCK>(L);
    const double ax = S->alpha[0], ay = S->alpha[1], ox = S->omega[0], oy = S->omega[1];
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    const bool plain = (s_hat == s);   // no preconditioner: one stream less
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * VEC_UNROLL) {
        double2 pi[VEC_UNROLL], sh[VEC_UNROLL], si[VEC_UNROLL], ti[VEC_UNROLL], rh[VEC_UNROLL], ui[VEC_UNROLL];
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            pi[q] = load_nt(p_hat + i);
            si[q] = load_nt(s + i);
            sh[q] = plain ? si[q] : load_nt(s_hat + i);
            ti[q] = load_nt(t + i);
            rh[q] = load_nt(r_hat + i);
            ui[q] = load_nt(u + i);
        }
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n) {
                double2 un = ui[q];
                un.x += ax * pi[q].x;   // BiCGStab.zig:329-331 (x += alpha * p_hat)
                un.y += ay * pi[q].y;
                un.x += ox * sh[q].x;   // BiCGStab.zig:352-354 (x += omega * s_hat)
                un.y += oy * sh[q].y;
                store_nt(u + i, un);
                const double2 ri = make_double2(si[q].x - ox * ti[q].x, si[q].y - oy * ti[q].y);   // BiCGStab.zig:356-358
                store_nt(r + i, ri);
                acc[0] += rh[q].x * ri.x;
                acc[1] += rh[q].y * ri.y;
                acc[2] += ri.x * ri.x;
                acc[3] += ri.y * ri.y;
            }
        }
    }
    block_partials<VEC_BLOCK, 4>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}
hipError_t launch_xr_update(const LazyScalars& S, double2* u, const double2* p_hat, const double2* s_hat, const double2* s, const double2* t, double2* r,
                            const double2* r_hat, int64_t n, double* partials, hipStream_t st) {
    hipLaunchKernelGGL(k_xr_update, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, S, u, p_hat, s_hat, s, t, r, r_hat, n, partials);
    return hipGetLastError();
}

// k_xr_update with s = r - alpha v formed on the fly (it was never stored: k_apply_vk<VK_S>): r is read and written in place
__global__ __launch_bounds__(VEC_BLOCK) void k_xr_update_vs(LazyScalars L, double2* __restrict__ u, const double2* __restrict__ p, const double2* __restrict__ v,
                                                            const double2* __restrict__ t, double2* __restrict__ r, const double2* __restrict__ r_hat, int64_t n,
                                                            double* partials) {
    const KrylovScalars* S = lazy_scalars<VEC_BLOCK>(L);
    const double ax = S->alpha[0], ay = S->alpha[1], ox = S->omega[0], oy = S->omega[1];
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * VEC_UNROLL) {
        double2 pi[VEC_UNROLL], ro[VEC_UNROLL], vi[VEC_UNROLL], ti[VEC_UNROLL], rh[VEC_UNROLL], ui[VEC_UNROLL];
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            pi[q] = load_nt(p + i);
            ro[q] = load_nt(r + i);
            vi[q] = load_nt(v + i);
            ti[q] = load_nt(t + i);
            rh[q] = load_nt(r_hat + i);
            ui[q] = load_nt(u + i);
        }
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n) {
                const double2 si = make_double2(ro[q].x - ax * vi[q].x, ro[q].y - ay * vi[q].y);   // BiCGStab.zig:325-327
                double2 un = ui[q];
                un.x += ax * pi[q].x;   // BiCGStab.zig:329-331 (x += alpha * p_hat)
                un.y += ay * pi[q].y;
                un.x += ox * si.x;      // BiCGStab.zig:352-354 (x += omega * s_hat)
                un.y += oy * si.y;
                store_nt(u + i, un);
                const double2 ri = make_double2(si.x - ox * ti[q].x, si.y - oy * ti[q].y);   // BiCGStab.zig:356-358
                store_nt(r + i, ri);
                acc[0] += rh[q].x * ri.x;
                acc[1] += rh[q].y * ri.y;
                acc[2] += ri.x * ri.x;
                acc[3] += ri.y * ri.y;
            }
        }
    }
    block_partials<VEC_BLOCK, 4>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}
hipError_t launch_xr_update_vs(const LazyScalars& S, double2* u, const double2* p, const double2* v, const double2* t, double2* r, const double2* r_hat, int64_t n,
                               double* partials, hipStream_t st) {
    hipLaunchKernelGGL(k_xr_update_vs, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, S, u, p, v, t, r, r_hat, n, partials);
    return hipGetLastError();
}

// K7 residual + copy-back (smooth.zig:112-153): sum (x_old - x_new)^2 per component, then the
// frozen field takes the new coordinates.
__global__ __launch_bounds__(VEC_BLOCK) void k_residual_copyback(double2* __restrict__ xk, const double2* __restrict__ u, int64_t n,
                                                                 double* partials) {
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * VEC_UNROLL) {
        double2 a[VEC_UNROLL], b[VEC_UNROLL];
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            a[q] = load_nt(xk + i);
            b[q] = load_nt(u + i);
        }
#pragma unroll
        for (int q = 0; q < VEC_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n) {
                const double dx = a[q].x - b[q].x, dy = a[q].y - b[q].y;
                acc[0] += dx * dx;
                acc[1] += dy * dy;
                store_nt(xk + i, b[q]);
            }
        }
    }
    block_partials<VEC_BLOCK, 2>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}
hipError_t launch_residual_copyback(double2* xk, const double2* u, int64_t n, double* partials, hipStream_t st) {
    hipLaunchKernelGGL(k_residual_copyback, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, xk, u, n, partials);
    return hipGetLastError();
}

// STREAM-style ceiling of this part at a given footprint, with the access pattern the vector kernels use (16 B per lane,
// non-temporal loads and stores, all loads of a trip issued before the first store): kind 0 = copy (b = a), 1 = triad
// (a = b + s c).  What `roofline.stream_ceiling_GBps` of bench.py reports beside the 8 TB/s specification.
template <int KIND>
__global__ __launch_bounds__(VEC_BLOCK) void k_stream(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s,
                                                      int64_t n) {
    constexpr int UN = 4;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * UN) {
        double2 x[UN], y[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            x[q] = load_nt(b + i);
            if (KIND == 1) y[q] = load_nt(c + i);
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n) store_nt(a + i, KIND == 1 ? make_double2(x[q].x + s * y[q].x, x[q].y + s * y[q].y) : x[q]);
        }
    }
}
hipError_t launch_stream(int kind, double2* a, const double2* b, const double2* c, double s, int64_t n, hipStream_t st) {
    // one trip per thread (grid = n / (256 x 4)): the fastest of the shapes tried in tools/ubench/stream.hip on this part
    const int64_t g = std::max<int64_t>(1, std::min<int64_t>((n + VEC_BLOCK * 4 - 1) / (VEC_BLOCK * 4), 1 << 20));
    if (kind == 1) hipLaunchKernelGGL(k_stream<1>, dim3(static_cast<unsigned>(g)), dim3(VEC_BLOCK), 0, st, a, b, c, s, n);
    else hipLaunchKernelGGL(k_stream<0>, dim3(static_cast<unsigned>(g)), dim3(VEC_BLOCK), 0, st, a, b, c, s, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(VEC_BLOCK) void k_gather_rows(const double2* __restrict__ src, const int32_t* __restrict__ ids, int64_t n,
                                                           double2* __restrict__ dst) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * VEC_BLOCK)
        dst[i] = src[ids[i]];
}
hipError_t launch_gather_rows(const double2* src, const int32_t* ids, int64_t n, double2* dst, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_rows, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, src, ids, n, dst);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_perimeter(const double2* __restrict__ in, double2* __restrict__ out, int ni, int nj) {
    const int k = blockIdx.x * 256 + threadIdx.x;   // 0..nj-1: row 0, nj..2nj-1: row ni-1, then columns 0 and nj-1 of rows 1..ni-2
    size_t id;
    if (k < nj) id = k;
    else if (k < 2 * nj) id = static_cast<size_t>(ni - 1) * nj + (k - nj);
    else {
        const int q = k - 2 * nj;
        const int i = 1 + (q >> 1);
        if (i > ni - 2) return;
        id = static_cast<size_t>(i) * nj + ((q & 1) ? nj - 1 : 0);
    }
    out[id] = in[id];
}
hipError_t launch_copy_perimeter(const double2* in, double2* out, int ni, int nj, hipStream_t st) {
    const int n = 2 * nj + 2 * (ni - 2);
    hipLaunchKernelGGL(k_copy_perimeter, dim3((n + 255) / 256), dim3(256), 0, st, in, out, ni, nj);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K1 tfi_blend -- reference tfi.zig:112-208 (boundary-blended linear TFI), same term order
// (types.addAll starts from (0,0) and adds left to right).  Write-only 16 B/node.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 d2_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 d2_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 d2_scale(double s, double2 v) { return make_double2(s * v.x, s * v.y); }

// u = nu / du, v = nv / dv, IEEE-rounded.  hipcc's f64 division is div_scale x2, rcp, 4 fma (Newton), mul, fma, div_fmas,
// div_fixup -- twice.  When du == dv and every operand is far from the ends of the exponent range (or a numerator is zero)
// the scalings are the identity and the fix-up passes the value through, so ONE reciprocal refinement serves both quotients
// and each costs mul + 2 fma more: the same bits (see recip_diag).  Anything else takes the plain divisions (wave-uniform).
__device__ __forceinline__ void div_pair(double nu, double nv, double du, double dv, double& u, double& v) {
    // Opposite edges with the same clustering (s2 == s1 or t2 == t1: every uniform block, BASELINE configs 2, 4, 5) make both
    // denominators exactly 1 - 0 = 1.0, and x / 1.0 == x bit for bit for every x: no division at all (wave-uniform test).
    if (__builtin_amdgcn_ballot_w64(!((du == 1.0) && (dv == 1.0))) == 0) {
        u = nu;
        v = nv;
        return;
    }
    const double au = fabs(nu), av = fabs(nv), ad = fabs(du);
    const bool plain = (du == dv) && (ad >= 0x1p-500) && (ad <= 0x1p500) && (au == 0.0 || (au >= 0x1p-500 && au <= 0x1p500)) &&
                       (av == 0.0 || (av >= 0x1p-500 && av <= 0x1p500));
    if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
        const double nd = -du;
        const double r0 = __builtin_amdgcn_rcp(du);
        const double e0 = fma(nd, r0, 1.0);
        const double r1 = fma(r0, e0, r0);
        const double e1 = fma(nd, r1, 1.0);
        const double r2 = fma(r1, e1, r1);
        const double qu = nu * r2, qv = nv * r2;
        u = fma(fma(nd, qu, nu), r2, qu);
        v = fma(fma(nd, qv, nv), r2, qv);
        return;
    }
    u = nu / du;
    v = nv / dv;
}

constexpr int TFI_ROWS = 8;   // rows per thread: the column's edge data (t1, t2, x_0j, x_nj) is loaded once per 8 nodes
__global__ __launch_bounds__(256) void k_tfi_block(double2* __restrict__ xy, int n, int m, const double2* __restrict__ x_i_min,
                                                   const double2* __restrict__ x_i_max, const double2* __restrict__ x_j_min,
                                                   const double2* __restrict__ x_j_max, const double* __restrict__ s1,
                                                   const double* __restrict__ s2, const double* __restrict__ t1,
                                                   const double* __restrict__ t2) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    // a wave = 64 consecutive columns of ONE group of rows: the row index is wave-uniform, so the row's edge data (s1, s2, x_i0,
    // x_im) comes through the scalar cache and all of a thread's rows are requested before the first is needed
    const int ib = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.y * 4 + threadIdx.y) * TFI_ROWS);
    if (j >= m || ib >= n) return;
    const double2 x_0_0 = x_i_min[0], x_n_0 = x_i_min[n - 1], x_0_m = x_j_min[m - 1], x_n_m = x_i_max[n - 1];
    const double t1_j = t1[j], t2_j = t2[j];
    const double2 x_0_j = x_j_min[j], x_n_j = x_j_max[j];
    double s1v[TFI_ROWS], s2v[TFI_ROWS];
    double2 xi0[TFI_ROWS], xim[TFI_ROWS];
#pragma unroll
    for (int k = 0; k < TFI_ROWS; ++k) {
        const int i = min(ib + k, n - 1);
        s1v[k] = s1[i];
        s2v[k] = s2[i];
        xi0[k] = x_i_min[i];
        xim[k] = x_i_max[i];
    }
    double2* dst = xy + static_cast<size_t>(ib) * m + j;
#pragma unroll
    for (int k = 0; k < TFI_ROWS; ++k) {
        if (ib + k >= n) break;   // scalar
        const double s1_i = s1v[k], s2_i = s2v[k];
        const double2 x_i_0 = xi0[k], x_i_m = xim[k];
        // tfi.zig:185-186: u = nu / (1 - (s2-s1)(t2-t1)), v = nv / (1 - (t2-t1)(s2-s1)) -- the two denominators are the same
        // number (one IEEE product, either order); both quotients correctly rounded like the reference's divisions
        double u, v;
        div_pair((1.0 - t1_j) * s1_i + t1_j * s2_i, (1.0 - s1_i) * t1_j + s1_i * t2_j, 1.0 - (s2_i - s1_i) * (t2_j - t1_j),
                 1.0 - (t2_j - t1_j) * (s2_i - s1_i), u, v);
        const double2 u_ij = d2_add(d2_scale(1.0 - u, x_0_j), d2_scale(u, x_n_j));
        const double2 v_ij = d2_add(d2_scale(1.0 - v, x_i_0), d2_scale(v, x_i_m));
        double2 uv = make_double2(0.0, 0.0);
        uv = d2_add(uv, d2_scale(u * v, x_n_m));
        uv = d2_add(uv, d2_scale(u * (1.0 - v), x_n_0));
        uv = d2_add(uv, d2_scale((1.0 - u) * v, x_0_m));
        uv = d2_add(uv, d2_scale((1.0 - u) * (1.0 - v), x_0_0));
        const double2 o = d2_sub(d2_add(u_ij, v_ij), uv);
        d2v ov;   // write-only stream of the whole block: non-temporal like the sweep kernels' stores
        ov.x = o.x;
        ov.y = o.y;
        __builtin_nontemporal_store(ov, reinterpret_cast<d2v*>(dst + static_cast<size_t>(k) * m));
    }
}
hipError_t launch_tfi_block(double2* xy, int ni, int nj, const double2* a, const double2* b, const double2* c, const double2* d,
                            const double* s1, const double* s2, const double* t1, const double* t2, hipStream_t st) {
    const dim3 block(64, 4), grid((nj + 63) / 64, (ni + 4 * TFI_ROWS - 1) / (4 * TFI_ROWS));
    hipLaunchKernelGGL(k_tfi_block, grid, block, 0, st, xy, ni, nj, a, b, c, d, s1, s2, t1, t2);
    return hipGetLastError();
}

// tfi.zig:19-67 (plain linear TFI, corners from the i edges)
__global__ __launch_bounds__(256) void k_tfi_linear2d(double2* __restrict__ xy, int n, int m, const double2* __restrict__ e_i_min,
                                                      const double2* __restrict__ e_i_max, const double2* __restrict__ e_j_min,
                                                      const double2* __restrict__ e_j_max) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= m || i >= n) return;
    const double2 c00 = e_i_min[0], c10 = e_i_min[n - 1], c01 = e_i_max[0], c11 = e_i_max[n - 1];
    const double xi = static_cast<double>(i) / static_cast<double>(n - 1);
    const double eta = static_cast<double>(j) / static_cast<double>(m - 1);
    const double2 u_ij = d2_add(d2_scale(1.0 - xi, e_j_min[j]), d2_scale(xi, e_j_max[j]));
    const double2 v_ij = d2_add(d2_scale(1.0 - eta, e_i_min[i]), d2_scale(eta, e_i_max[i]));
    double2 uv = make_double2(0.0, 0.0);
    uv = d2_add(uv, d2_scale(xi * eta, c11));
    uv = d2_add(uv, d2_scale(xi * (1.0 - eta), c10));
    uv = d2_add(uv, d2_scale((1.0 - xi) * eta, c01));
    uv = d2_add(uv, d2_scale((1.0 - xi) * (1.0 - eta), c00));
    double2 res = make_double2(0.0, 0.0);
    res = d2_add(res, u_ij);
    res = d2_add(res, v_ij);
    res = d2_add(res, make_double2(-uv.x, -uv.y));
    xy[static_cast<size_t>(i) * m + j] = res;
}
hipError_t launch_tfi_linear2d(double2* xy, int ni, int nj, const double2* a, const double2* b, const double2* c, const double2* d,
                               hipStream_t st) {
    const dim3 block(64, 4), grid((nj + 63) / 64, (ni + 3) / 4);
    hipLaunchKernelGGL(k_tfi_linear2d, grid, block, 0, st, xy, ni, nj, a, b, c, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K6 white control function -- wall_control_function.zig:70-473 (hard-coded to blocks 0,1 and
// connection 0 like the reference).  wall: one thread per wall node writes (P,Q) at j = 0;
// le: the leading-edge node of block 0 across connection 0; blend: factor = 1 - j/(nj-1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 eq610(double x_xi, double y_xi, double x_xi2, double y_xi2, double x_eta, double y_eta,
                                         double x_eta2, double y_eta2) {
    const double g11 = x_xi * x_xi + y_xi * y_xi;
    const double g22 = x_eta * x_eta + y_eta * y_eta;
    const double p = -(x_xi * x_xi2 + y_xi * y_xi2) / g11 - (x_xi * x_eta2 + y_xi * y_eta2) / g22;
    const double q = -(x_eta * x_eta2 + y_eta * y_eta2) / g22 - (x_eta * x_xi2 + y_eta * y_xi2) / g11;
    return make_double2(p, q);
}
__device__ __forceinline__ double2 white_delta(double x_xi, double y_xi, double x_eta, double y_eta, double ds_target,
                                               double theta_target) {   // wall_control_function.zig:293-304
    const double g11 = x_xi * x_xi + y_xi * y_xi;
    const double g12 = x_xi * x_eta + y_xi * y_eta;
    const double g22 = x_eta * x_eta + y_eta * y_eta;
    const double ds = sqrt(g22);
    const double theta = tm_refmath::acos(g12 / sqrt(g11 * g22));   // the reference's libm, not ocml's (tm_refmath.h)
    const double delta_ds = ds_target - ds;
    const double delta_theta = theta_target - theta;
    return make_double2(-tm_refmath::atan2(delta_theta, theta_target), tm_refmath::atan2(delta_ds, ds_target));
}

__global__ void k_white_wall(const double2* __restrict__ d, double2* __restrict__ pq, int ni, int nj, int update, double ds_target,
                             double theta_target) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ni) return;
    const size_t id = static_cast<size_t>(i) * nj;
    const double2 x0 = d[id], x1 = d[id + 1];
    const double x_eta = -x0.x + x1.x, y_eta = -x0.y + x1.y;   // forward eta difference
    double x_xi, y_xi, x_xi2 = 0.0, y_xi2 = 0.0;
    if (i == 0) {   // forward xi (:86-90, 339-341)
        const double2 a = d[id + nj];
        x_xi = -x0.x + a.x;
        y_xi = -x0.y + a.y;
        if (!update) {
            const double2 b = d[id + 2 * static_cast<size_t>(nj)];
            x_xi2 = x0.x - 2 * a.x + b.x;
            y_xi2 = x0.y - 2 * a.y + b.y;
        }
    } else if (i == ni - 1) {   // backward xi (:165-170, 376-378)
        const double2 a = d[id - nj];
        x_xi = x0.x - a.x;
        y_xi = x0.y - a.y;
        if (!update) {
            const double2 b = d[id - 2 * static_cast<size_t>(nj)];
            x_xi2 = x0.x - 2 * a.x + b.x;
            y_xi2 = x0.y - 2 * a.y + b.y;
        }
    } else {   // central xi (:124-128, 357-359)
        const double2 xp = d[id + nj], xm = d[id - nj];
        x_xi = 0.5 * (xp.x - xm.x);
        y_xi = 0.5 * (xp.y - xm.y);
        x_xi2 = xp.x - 2 * x0.x + xm.x;
        y_xi2 = xp.y - 2 * x0.y + xm.y;
    }
    if (!update) {
        const double2 x2 = d[id + 2];
        const double x_eta2 = x0.x - 2 * x1.x + x2.x, y_eta2 = x0.y - 2 * x1.y + x2.y;
        pq[id] = eq610(x_xi, y_xi, x_xi2, y_xi2, x_eta, y_eta, x_eta2, y_eta2);
    } else {
        const double2 dl = white_delta(x_xi, y_xi, x_eta, y_eta, ds_target, theta_target);
        double2 v = pq[id];
        v.x += 0.1 * dl.x;
        v.y += 0.1 * dl.y;
        pq[id] = v;
    }
}
__global__ void k_white_le(WhiteArgs w, int update) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double2 xij = w.x0[w.le_p0];
    const double2 xip1 = w.x0[w.le_p0 + w.le_fi0];
    const double2 xim1 = w.x1[w.le_p1 + w.le_fi1];
    const double2 xjp1 = w.x0[w.le_p0 + w.le_dir0];
    const double x_eta = -xij.x + xjp1.x, y_eta = -xij.y + xjp1.y;
    if (!update) {   // wall_control_function.zig:235-259
        const double2 xjp2 = w.x0[w.le_p0 + 2 * w.le_dir0];
        const double x_xi = 0.5 * (xip1.x - xim1.x), y_xi = 0.5 * (xip1.y - xim1.y);
        const double x_xi2 = xip1.x - 2 * xij.x + xim1.x, y_xi2 = xip1.y - 2 * xij.y + xim1.y;
        const double x_eta2 = xij.x - 2 * xjp1.x + xjp2.x, y_eta2 = xij.y - 2 * xjp1.y + xjp2.y;
        w.pq0[0] = eq610(x_xi, y_xi, x_xi2, y_xi2, x_eta, y_eta, x_eta2, y_eta2);
    } else {   // :423-452, negated xi difference (:428-431)
        const double x_xi = -0.5 * (xip1.x - xim1.x), y_xi = -0.5 * (xip1.y - xim1.y);
        const double2 dl = white_delta(x_xi, y_xi, x_eta, y_eta, w.ds_target, w.theta_target);
        double2 v = w.pq0[0];
        v.x += 0.1 * dl.x;
        v.y += 0.1 * dl.y;
        w.pq0[0] = v;
    }
}
__global__ void k_white_blend(double2* __restrict__ pq, int ni, int nj) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < 1 || j >= nj || i >= ni) return;
    const double2 wall = pq[static_cast<size_t>(i) * nj];
    const double factor = 1 - static_cast<double>(j) / (static_cast<double>(nj) - 1);   // :108
    pq[static_cast<size_t>(i) * nj + j] = make_double2(factor * wall.x, factor * wall.y);
}
hipError_t launch_white(const WhiteArgs& w, int update, hipStream_t st) {
    hipLaunchKernelGGL(k_white_wall, dim3((w.ni0 + 63) / 64), dim3(64), 0, st, w.x0, w.pq0, w.ni0, w.nj0, update, w.ds_target, w.theta_target);
    hipLaunchKernelGGL(k_white_wall, dim3((w.ni1 + 63) / 64), dim3(64), 0, st, w.x1, w.pq1, w.ni1, w.nj1, update, w.ds_target, w.theta_target);
    hipLaunchKernelGGL(k_white_le, dim3(1), dim3(64), 0, st, w, update);
    hipLaunchKernelGGL(k_white_blend, dim3((w.nj0 + 63) / 64, w.ni0), dim3(64), 0, st, w.pq0, w.ni0, w.nj0);
    hipLaunchKernelGGL(k_white_blend, dim3((w.nj1 + 63) / 64, w.ni1), dim3(64), 0, st, w.pq1, w.ni1, w.nj1);
    return hipGetLastError();
}


// diagnostic (tm_white_math_probe): the two libm functions of the White control function, evaluated on the device
__global__ __launch_bounds__(256) void k_debug_white_math(const double* __restrict__ x, const double* __restrict__ y, uint64_t n, double* __restrict__ out_acos,
                                                          double* __restrict__ out_atan2) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    out_acos[i] = tm_refmath::acos(x[i]);
    out_atan2[i] = tm_refmath::atan2(y[i], x[i]);
}
hipError_t launch_debug_white_math(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_debug_white_math, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, x, y, n, out_acos, out_atan2);
    return hipGetLastError();
}

}  // namespace tmh
