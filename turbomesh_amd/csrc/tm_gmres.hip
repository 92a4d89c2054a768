// GMRES(m) on the device -- the reference's other Krylov solver (reference src/core/smoothing/GMRES.zig:300-423: left-preconditioned
// restarted GMRES, modified Gram-Schmidt Arnoldi, Givens rotations :510-524, back substitution :394-409), served as an inner strategy of
// the hip solver (TM_INNER_GMRES) with the diagonal preconditioner (GMRES.zig:425-431): z = D^-1 (A v) is exactly K2's MODE_SCALED, so the
// operator application is the same matrix-free kernel the BiCGStab path uses.  ILU(0) (GMRES.zig:199-298) is a sequential recurrence and
// stays on the CPU side (oracle only).
//
// Both coordinate components advance together (double2 vectors, independent scalars per component -- the reference solves the x- and the
// y-system one after the other with the same code, smooth.zig / solver.zig:69-78); a component that has converged inside a restart cycle
// freezes its column count, the other goes on, and the update x += V y uses each component's own columns.
//
// Everything that decides the iteration lives in device memory (GmresScalars): Hessenberg columns, rotations, the rotated right-hand side,
// tolerances, flags.  The host enqueues and polls the flags every `check_every` columns.
//
// Deviation from the reference, the same as on the BiCGStab path: the stop test is on the SCALED residual,
// ||D^-1 (b - A x)||_2 <= max(atol, rtol ||D^-1 b||_2) -- which with the diagonal as left preconditioner is GMRES's own residual norm
// |g_{j+1}| -- instead of the reference's max(1e-8, 1e-6 ||b||) on unscaled b.
#include "tm_devutil.hpp"

namespace tmh {

// ---- vector kernels: 16 B per lane, two elements per trip, owned rows only
namespace {
constexpr int GV_UNROLL = 2;

// w' = w - h_prev * v_prev (stored, has_prev) ; partials: w' . v_next (x, y).  One MGS step = ONE pass: the subtraction of the previous
// projection and the next inner product travel together (GMRES.zig:338-345 reads and writes z once per basis vector as well).
template <bool HAS_PREV, bool HAS_NEXT>
__global__ __launch_bounds__(VEC_BLOCK) void k_gm_mgs(double2* __restrict__ w, const double2* __restrict__ v_prev, const double2* __restrict__ v_next,
                                                      const double* __restrict__ red_prev, GmresScalars* __restrict__ G, int row_prev, int64_t n,
                                                      double* __restrict__ partials) {
    double hx = 0.0, hy = 0.0;
    if (HAS_PREV) {
        hx = G->done[0] ? 0.0 : red_prev[0];   // a finished component's columns are not used any more: leave its w alone
        hy = G->done[1] ? 0.0 : red_prev[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {   // h_{row_prev, j} (GMRES.zig:341)
            const int j = G->j;
            if (!G->done[0]) G->H[0][row_prev + (GMRES_M + 1) * j] = hx;
            if (!G->done[1]) G->H[1][row_prev + (GMRES_M + 1) * j] = hy;
        }
    }
    double acc[MAX_PARTIALS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i0 < n; i0 += stride * GV_UNROLL) {
        double2 wi[GV_UNROLL], pi[GV_UNROLL], ni[GV_UNROLL];
#pragma unroll
        for (int q = 0; q < GV_UNROLL; ++q) {
            const int64_t i = min(i0 + q * stride, n - 1);
            wi[q] = w[i];
            if (HAS_PREV) pi[q] = v_prev[i];
            if (HAS_NEXT) ni[q] = v_next[i];
        }
#pragma unroll
        for (int q = 0; q < GV_UNROLL; ++q) {
            const int64_t i = i0 + q * stride;
            if (i >= n) continue;
            double2 z = wi[q];
            if (HAS_PREV) {
                z.x -= hx * pi[q].x;   // z[k] -= h_ij * vi[k], GMRES.zig:343
                z.y -= hy * pi[q].y;
                w[i] = z;
            }
            if (HAS_NEXT) {
                acc[0] += z.x * ni[q].x;   // dot(z, vi), GMRES.zig:340
                acc[1] += z.y * ni[q].y;
            } else {
                acc[0] += z.x * z.x;       // norm(z)^2, GMRES.zig:347
                acc[1] += z.y * z.y;
            }
        }
    }
    block_partials<VEC_BLOCK, 2>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}

// dst = src / denom[c] per component -- the reference's division, bit for bit.  scale_skip[c] = 1: the component keeps what dst holds
// (GMRES.zig:350-355: v_{j+1} is only written when h_next > breakdown_eps); 2: it is cleared (a component that needs no iteration at all).
__global__ __launch_bounds__(VEC_BLOCK) void k_gm_divide(double2* __restrict__ dst, const double2* __restrict__ src, const GmresScalars* __restrict__ G, int64_t n) {
    const double dx = G->scale[0], dy = G->scale[1];
    const bool kx = G->scale_skip[0] != 0, ky = G->scale_skip[1] != 0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i < n; i += stride) {
        const double2 s = src[i];
        double2 d = dst[i];
        if (!kx) d.x = s.x / dx;   // v0[i] = z[i] / beta, vnext[k] = z[k] / h_next (GMRES.zig:322, 353)
        if (!ky) d.y = s.y / dy;
        if (G->scale_skip[0] == 2) d.x = 0.0;
        if (G->scale_skip[1] == 2) d.y = 0.0;
        dst[i] = d;
    }
}

// u += sum_{i < cols_used[c]} y[c][i] v_i  (GMRES.zig:411-417: basis vector by basis vector, in this order)
__global__ __launch_bounds__(VEC_BLOCK) void k_gm_update(double2* __restrict__ u, const double2* __restrict__ V, int64_t ld, const GmresScalars* __restrict__ G, int64_t n) {
    const int cx = G->cols_used[0], cy = G->cols_used[1];
    const int cmax = cx > cy ? cx : cy;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * VEC_BLOCK;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(VEC_BLOCK) + threadIdx.x; i < n; i += stride) {
        double2 x = u[i];
        for (int k = 0; k < cmax; ++k) {
            const double2 v = V[static_cast<int64_t>(k) * ld + i];
            if (k < cx) x.x += G->y[0][k] * v.x;
            if (k < cy) x.y += G->y[1][k] * v.y;
        }
        u[i] = x;
    }
}

// ---- scalar kernels: one thread per component
// tolerance from ||D^-1 b||^2 (red[0..1]): tol = max(atol, rtol ||D^-1 b||)
__global__ void k_gm_tol(GmresScalars* G, const double* red, double rtol, double atol) {
    const int c = threadIdx.x;
    if (c >= 2) return;
    G->tol[c] = fmax(atol, rtol * sqrt(red[c]));
    G->tol_initial[c] = 0;
    if (rtol < 0.0) {   // TM_OPT_RTOL_INITIAL: relative to the initial residual of the solve, known at the first k_gm_begin
        G->tol[c] = atol;
        G->tol_initial[c] = 1;
        G->rtol_initial = -rtol;
    }
    G->cycle = 0;
}
// start of a restart cycle: red[0..1] = ||D^-1 (b - A x)||^2 (GMRES.zig:312-336)
__global__ void k_gm_begin(GmresScalars* G, const double* red) {
    const int c = threadIdx.x;
    if (c >= 2) return;
    const double beta = sqrt(red[c]);
    G->beta[c] = beta;
    if (G->cycle == 0) {
        G->rr0[c] = red[c];
        if (G->tol_initial[c]) G->tol[c] = fmax(G->tol[c], G->rtol_initial * beta);
    }
    G->resid[c] = beta;
    G->cols_used[c] = 0;
    G->done[c] = beta <= G->tol[c] ? 1 : 0;   // if (beta <= tol) return
    G->scale[c] = beta;
    G->scale_skip[c] = G->done[c] ? 2 : 0;    // a finished component gets a zero v0 instead of 0 / 0
    for (int k = 0; k < (GMRES_M + 1) * GMRES_M; ++k) G->H[c][k] = 0.0;
    for (int k = 0; k < GMRES_M; ++k) G->cs[c][k] = G->sn[c][k] = G->y[c][k] = 0.0;
    for (int k = 0; k <= GMRES_M; ++k) G->g[c][k] = 0.0;
    G->g[c][0] = beta;
    if (c == 0) {
        G->j = 0;
        G->cycle += 1;
    }
}
// end of column j: red[0..1] = ||z||^2 after the projections; h_jj travelled in red_h (the last projection)
__global__ void k_gm_column(GmresScalars* G, const double* red) {
    const int c = threadIdx.x;
    const int j = G->j;
    if (c < 2 && !G->done[c]) {
        double* H = G->H[c];
        auto h = [&](int row, int col) -> double& { return H[row + (GMRES_M + 1) * col]; };
        const double h_next = sqrt(red[c]);
        h(j + 1, j) = h_next;
        G->scale[c] = h_next;
        G->scale_skip[c] = h_next > 1e-30 ? 0 : 1;                     // breakdown_eps, GMRES.zig:301, 350
        for (int i = 0; i < j; ++i) {                                   // previous rotations on the new column, GMRES.zig:357-363
            const double h_i = h(i, j), h_ip1 = h(i + 1, j);
            const double temp = G->cs[c][i] * h_i + G->sn[c][i] * h_ip1;
            h(i + 1, j) = -G->sn[c][i] * h_i + G->cs[c][i] * h_ip1;
            h(i, j) = temp;
        }
        const double a = h(j, j), b = h(j + 1, j);                      // computeGivens, GMRES.zig:510-524
        double rc, rs, rr;
        if (b == 0.0) {
            rc = 1.0;
            rs = 0.0;
            rr = a;
        } else if (fabs(b) > fabs(a)) {
            const double t = a / b;
            const double s = 1.0 / sqrt(1.0 + t * t);
            rc = s * t;
            rs = s;
            rr = b / s;
        } else {
            const double t = b / a;
            const double cc = 1.0 / sqrt(1.0 + t * t);
            rc = cc;
            rs = cc * t;
            rr = a / cc;
        }
        G->cs[c][j] = rc;
        G->sn[c][j] = rs;
        h(j, j) = rr;
        h(j + 1, j) = 0.0;
        const double g_j = G->g[c][j], g_jp1 = G->g[c][j + 1];
        G->g[c][j] = rc * g_j + rs * g_jp1;
        G->g[c][j + 1] = -rs * g_j + rc * g_jp1;
        G->resid[c] = fabs(G->g[c][j + 1]);
        G->cols_used[c] = j + 1;
        if (G->resid[c] <= G->tol[c]) G->done[c] = 1;                   // converged: this component's cycle ends here
    } else if (c < 2) {
        G->scale_skip[c] = 1;   // finished earlier in this cycle: its basis is not extended
    }
    __syncthreads();
    if (c == 0) G->j = j + 1;
}
// y from the triangular system (GMRES.zig:394-409)
__global__ void k_gm_backsub(GmresScalars* G) {
    const int c = threadIdx.x;
    if (c >= 2) return;
    const int n = G->cols_used[c];
    const double* H = G->H[c];
    for (int k = 0; k < GMRES_M; ++k) G->y[c][k] = 0.0;
    for (int idx = n; idx > 0;) {
        idx -= 1;
        double sum = G->g[c][idx];
        for (int k = idx + 1; k < n; ++k) sum -= H[idx + (GMRES_M + 1) * k] * G->y[c][k];
        const double h_ii = H[idx + (GMRES_M + 1) * idx];
        if (h_ii == 0.0) break;
        G->y[c][idx] = sum / h_ii;
    }
}
}  // namespace

hipError_t launch_gm_mgs(double2* w, const double2* v_prev, const double2* v_next, const double* red_prev, GmresScalars* G, int row_prev, int64_t n,
                         double* partials, hipStream_t st) {
    const int g = vec_nwg(n);
    if (v_prev && v_next) hipLaunchKernelGGL((k_gm_mgs<true, true>), dim3(g), dim3(VEC_BLOCK), 0, st, w, v_prev, v_next, red_prev, G, row_prev, n, partials);
    else if (v_prev) hipLaunchKernelGGL((k_gm_mgs<true, false>), dim3(g), dim3(VEC_BLOCK), 0, st, w, v_prev, v_next, red_prev, G, row_prev, n, partials);
    else hipLaunchKernelGGL((k_gm_mgs<false, true>), dim3(g), dim3(VEC_BLOCK), 0, st, w, v_prev, v_next, red_prev, G, row_prev, n, partials);
    return hipGetLastError();
}
hipError_t launch_gm_divide(double2* dst, const double2* src, const GmresScalars* G, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_divide, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, dst, src, G, n);
    return hipGetLastError();
}
hipError_t launch_gm_update(double2* u, const double2* V, int64_t ld, const GmresScalars* G, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_update, dim3(vec_nwg(n)), dim3(VEC_BLOCK), 0, st, u, V, ld, G, n);
    return hipGetLastError();
}
hipError_t launch_gm_tol(GmresScalars* G, const double* red, double rtol, double atol, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_tol, dim3(1), dim3(64), 0, st, G, red, rtol, atol);
    return hipGetLastError();
}
hipError_t launch_gm_begin(GmresScalars* G, const double* red, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_begin, dim3(1), dim3(64), 0, st, G, red);
    return hipGetLastError();
}
hipError_t launch_gm_column(GmresScalars* G, const double* red, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_column, dim3(1), dim3(64), 0, st, G, red);
    return hipGetLastError();
}
hipError_t launch_gm_backsub(GmresScalars* G, hipStream_t st) {
    hipLaunchKernelGGL(k_gm_backsub, dim3(1), dim3(64), 0, st, G);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The multigrid preconditioner's perimeter step (Smoother::precondition): out = in - h + out on the perimeter nodes of a block, nothing else
// touched -- with h = the perimeter rows applied to out, a Jacobi sweep on the (unit-diagonal) perimeter system; out = in - h when out was zero there.  (Here, not beside k_copy_perimeter whose indexing it shares: profiles/traffic.json is keyed to the text of tm_kernels.hip.)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_perimeter_sub(const double2* __restrict__ in, const double2* __restrict__ h, double2* __restrict__ out, int ni, int nj) {
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
    const double2 a = in[id], b = h[id], c = out[id];   // (out is zero there in the first pass: x + 0 == x to the bit)
    out[id] = make_double2((a.x - b.x) + c.x, (a.y - b.y) + c.y);
}
hipError_t launch_perimeter_sub(const double2* in, const double2* h, double2* out, int ni, int nj, hipStream_t st) {
    const int n = 2 * nj + 2 * (ni - 2);
    hipLaunchKernelGGL(k_perimeter_sub, dim3((n + 255) / 256), dim3(256), 0, st, in, h, out, ni, nj);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The other half of the preconditioner's perimeter treatment: the perimeter values f_p as DIRICHLET DATA of the block's cycle, i.e. the
// right-hand side of the first interior ring becomes f_I - (D^-1 A)_Ip f_p for the duration of the cycle (saved, changed, restored to the bit
// by k_ring_restore -- the vector is the Krylov method's own p or s).  A thread per ring node: the row's nine coefficients as
// stencil_coefs forms them (tm_kernels.hip; smooth.zig:171-216), the sum over the neighbours that lie on the perimeter, scaled by
// the diagonal like MODE_SCALED.  Any fixed linear operator would do here: this shapes the preconditioner, not the system.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ring_node(int k, int ni, int nj, int& i, int& j) {   // 2 (nj - 2) + 2 (ni - 4) nodes, ni, nj >= 5
    const int w = nj - 2;
    if (k < w) { i = 1; j = 1 + k; return true; }
    if (k < 2 * w) { i = ni - 2; j = 1 + (k - w); return true; }
    const int q = k - 2 * w;
    i = 2 + (q >> 1);
    j = (q & 1) ? nj - 2 : 1;
    return i <= ni - 3;
}
template <bool HAS_PQ>
__global__ __launch_bounds__(256) void k_ring_dirichlet(double2* __restrict__ f, const double2* __restrict__ X, const double2* __restrict__ PQ, double2* __restrict__ save,
                                                        int ni, int nj) {
    int i, j;
    if (!ring_node(blockIdx.x * 256 + threadIdx.x, ni, nj, i, j)) return;
    const size_t id = static_cast<size_t>(i) * nj + j;
    const double2 xm = X[id - nj], xp = X[id + nj], xl = X[id - 1], xr = X[id + 1];
    const double x_xi = 0.5 * (xp.x - xm.x), y_xi = 0.5 * (xp.y - xm.y), x_eta = 0.5 * (xr.x - xl.x), y_eta = 0.5 * (xr.y - xl.y);
    const double g22 = x_eta * x_eta + y_eta * y_eta, g12 = x_xi * x_eta + y_xi * y_eta, g11 = x_xi * x_xi + y_xi * y_xi;
    const double diag = -2.0 * g22 - 2.0 * g11;
    double P = 0.0, Q = 0.0;
    if (HAS_PQ) { const double2 pq = PQ[id]; P = pq.x; Q = pq.y; }
    const double c_ip = g22 * (1.0 + 0.5 * P), c_im = g22 * (1.0 - 0.5 * P), c_jp = g11 * (1.0 + 0.5 * Q), c_jm = g11 * (1.0 - 0.5 * Q);
    const double c_pp = -0.5 * g12, c_pm = 0.5 * g12, c_mp = 0.5 * g12, c_mm = -0.5 * g12;
    const bool top = i == 1, bottom = i == ni - 2, left = j == 1, right = j == nj - 2;   // the neighbours beyond are perimeter nodes
    double sx = 0.0, sy = 0.0;
    auto add = [&](double c, size_t at) { const double2 v = f[at]; sx += c * v.x; sy += c * v.y; };
    if (top) add(c_im, id - nj);
    if (bottom) add(c_ip, id + nj);
    if (left) add(c_jm, id - 1);
    if (right) add(c_jp, id + 1);
    if (top || left) add(c_mm, id - nj - 1);
    if (top || right) add(c_mp, id - nj + 1);
    if (bottom || left) add(c_pm, id + nj - 1);
    if (bottom || right) add(c_pp, id + nj + 1);
    const double dinv = (diag == 0.0) ? 1.0 : 1.0 / diag;
    const double2 v = f[id];
    save[id] = v;
    f[id] = make_double2(v.x - sx * dinv, v.y - sy * dinv);
}
__global__ __launch_bounds__(256) void k_ring_restore(double2* __restrict__ f, const double2* __restrict__ save, int ni, int nj) {
    int i, j;
    if (!ring_node(blockIdx.x * 256 + threadIdx.x, ni, nj, i, j)) return;
    const size_t id = static_cast<size_t>(i) * nj + j;
    f[id] = save[id];
}
hipError_t launch_ring_dirichlet(double2* f, const double2* X, const double2* PQ, double2* save, int ni, int nj, hipStream_t st) {
    if (ni < 5 || nj < 5) return hipSuccess;
    const int n = 2 * (nj - 2) + 2 * (ni - 4);
    if (PQ) hipLaunchKernelGGL(k_ring_dirichlet<true>, dim3((n + 255) / 256), dim3(256), 0, st, f, X, PQ, save, ni, nj);
    else hipLaunchKernelGGL(k_ring_dirichlet<false>, dim3((n + 255) / 256), dim3(256), 0, st, f, X, PQ, save, ni, nj);
    return hipGetLastError();
}
hipError_t launch_ring_restore(double2* f, const double2* save, int ni, int nj, hipStream_t st) {
    if (ni < 5 || nj < 5) return hipSuccess;
    const int n = 2 * (nj - 2) + 2 * (ni - 4);
    hipLaunchKernelGGL(k_ring_restore, dim3((n + 255) / 256), dim3(256), 0, st, f, save, ni, nj);
    return hipGetLastError();
}

}  // namespace tmh
