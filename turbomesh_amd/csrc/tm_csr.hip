// Seam 2 -- the reference's linear-solver slot (reference src/core/smoothing/solver.zig:40-93): a backend receives the
// ASSEMBLED system `RowCompressedMatrixSystem2d` (smooth.zig:277-307: lhs_p, lhs_i, lhs_values, rhs_x, rhs_y, x_new, y_new)
// and solves the x- and the y-system, calling system.fillXSpecific() / fillYSpecific() before the respective solve
// (umfpack.zig:18-24) -- the two systems share the pattern and differ only in the two entries of each `sliding_circ` row
// (smooth.zig:1115-1165).  tm_csr_solve is that backend on the MI355X: caller-assembled CSR in, both components solved
// TOGETHER as double2 vectors with the same device-resident BiCGStab as the matrix-free path (recurrences of
// BiCGStab.zig:279-370 on the row-equilibrated system D^-1 A x = D^-1 b, scale-aware stop test, restart on breakdown).
// Not the performance path: the matrix travels over PCIe per call and the mat-vec streams 12 B per non-zero; it exists so
// that a Zig `Solver` arm needs no change to smooth.mesh, and so that the tests have an operator check that never touches
// the matrix-free kernels.
#include "tm_api_util.hpp"
#include "tm_devutil.hpp"
#include <chrono>
#include <cstring>

namespace tmh {

namespace {

struct CsrDev {
    int n;
    const int32_t* p;
    const int32_t* i;
    const double* vx;   // values of the x-system
    const double* vy;   // values of the y-system (== vx when the caller passes one array)
    const double2* dinv;
};

// 1 / a_ii per row and component; a missing or zero diagonal scales by 1 (BiCGStab.zig:155-175)
__global__ __launch_bounds__(256) void k_csr_dinv(int n, const int32_t* __restrict__ p, const int32_t* __restrict__ ci, const double* __restrict__ vx,
                                                  const double* __restrict__ vy, double2* __restrict__ dinv) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    double dx = 0.0, dy = 0.0;
    for (int k = p[row]; k < p[row + 1]; ++k)
        if (ci[k] == row) {
            dx = vx[k];
            dy = vy[k];
            break;
        }
    dinv[row] = make_double2(dx == 0.0 ? 1.0 : 1.0 / dx, dy == 0.0 ? 1.0 : 1.0 / dy);
}

// one thread per row (<= 9 non-zeros in the reference's systems).  RESID: out = D^-1 (b - A in), else out = D^-1 A in;
// fused partial dot products like K2 (tm_kernels.h DotMode)
template <bool RESID, int DOT>
__global__ __launch_bounds__(256) void k_csr_apply(CsrDev A, const double2* __restrict__ in, const double2* __restrict__ b, const double2* __restrict__ aux,
                                                   double2* __restrict__ out, double* partials) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (row < A.n) {
        double sx = 0.0, sy = 0.0;
        for (int k = A.p[row]; k < A.p[row + 1]; ++k) {   // BiCGStab.zig:424-435, both components
            const double2 w = in[A.i[k]];
            sx += A.vx[k] * w.x;
            sy += A.vy[k] * w.y;
        }
        const double2 d = A.dinv[row];
        double2 o;
        if (RESID) {
            const double2 rhs = b[row];
            o = make_double2(rhs.x * d.x - sx * d.x, rhs.y * d.y - sy * d.y);
        } else {
            o = make_double2(sx * d.x, sy * d.y);
        }
        out[row] = o;
        if (DOT == DOT_AUX) {
            const double2 a = aux[row];
            acc[0] = a.x * o.x;
            acc[1] = a.y * o.y;
        } else if (DOT == DOT_IN) {
            const double2 w = in[row];
            acc[0] = w.x * o.x;
            acc[1] = w.y * o.y;
            acc[2] = o.x * o.x;
            acc[3] = o.y * o.y;
        } else if (DOT == DOT_OUT2) {
            acc[0] = o.x * o.x;
            acc[1] = o.y * o.y;
        }
    }
    if (DOT != DOT_NONE) block_partials<256>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}

// partials of ||D^-1 b||^2 per component (the stop test's reference norm)
__global__ __launch_bounds__(256) void k_csr_bnorm(int n, const double2* __restrict__ b, const double2* __restrict__ dinv, double* partials) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (row < n) {
        const double2 v = b[row], d = dinv[row];
        acc[0] = (v.x * d.x) * (v.x * d.x);
        acc[1] = (v.y * d.y) * (v.y * d.y);
    }
    block_partials<256>(acc, partials + static_cast<size_t>(blockIdx.x) * MAX_PARTIALS);
}

__global__ __launch_bounds__(256) void k_interleave(int n, const double* __restrict__ a, const double* __restrict__ b, double2* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_double2(a[i], b[i]);
}
__global__ __launch_bounds__(256) void k_deinterleave(int n, const double2* __restrict__ in, double* __restrict__ a, double* __restrict__ b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double2 v = in[i];
        a[i] = v.x;
        b[i] = v.y;
    }
}

struct Dev {   // RAII device buffer
    void* p = nullptr;
    explicit Dev(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed (" + std::to_string(bytes) + " bytes)");
    }
    ~Dev() { (void)hipFree(p); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

}  // namespace

}  // namespace tmh

using namespace tmh;

extern "C" int tm_csr_solve(uint64_t n64, const int32_t* Ap, const int32_t* Ai, const double* Ax_x, const double* Ax_y, const double* bx,
                            const double* by, double* x, double* y, const tm_solver_opt* opt_in, tm_stats* stats) {
    return guarded([&]() {
        const auto t0 = std::chrono::steady_clock::now();
        if (!Ap || !Ai || !Ax_x || !bx || !by || !x || !y) throw TmError(TM_E_ARG, "null argument");
        if (n64 == 0 || n64 >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "system size out of range");
        if (opt_in && opt_in->tag != TM_SOLVER_HIP)
            throw TmError(TM_E_UNSUPPORTED, "ExternalSolverNotEnabled: libtm_hip serves only solver tag `hip`");
        const int n = static_cast<int>(n64);
        if (Ap[0] != 0 || Ap[n] < 0) throw TmError(TM_E_ARG, "InvalidMatrix: row pointers must start at 0");
        const size_t nnz = static_cast<size_t>(Ap[n]);
        for (int r = 0; r < n; ++r)
            if (Ap[r + 1] < Ap[r]) throw TmError(TM_E_ARG, "InvalidMatrix: row pointers must not decrease");
        for (size_t k = 0; k < nnz; ++k)
            if (Ai[k] < 0 || Ai[k] >= n) throw TmError(TM_E_ARG, "InvalidMatrix: column index out of range");
        tm_solver_opt opt;
        std::memset(&opt, 0, sizeof(opt));
        if (opt_in) opt = *opt_in;
        // 1e-14 at every size: the size-aware default of the matrix-free path belongs to ITS operator (the frozen Winslow system, whose
        // conditioning grows with the node count); this entry point takes any matrix the caller assembled
        if (!(opt.rtol > 0)) opt.rtol = 1e-14;
        if (!(opt.atol > 0)) opt.atol = 0.0;
        if (opt.max_inner == 0) opt.max_inner = default_max_inner(static_cast<double>(n));
        if (opt.check_every == 0) opt.check_every = 8;

        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) throw TmError(TM_E_HIP, std::string("libtm_hip is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);

        const bool two = Ax_y != nullptr && Ax_y != Ax_x;
        const size_t vb = sizeof(double2) * static_cast<size_t>(n);
        Dev d_p(sizeof(int32_t) * (static_cast<size_t>(n) + 1)), d_i(sizeof(int32_t) * nnz), d_vx(sizeof(double) * nnz), d_vy(two ? sizeof(double) * nnz : 0);
        Dev d_dinv(vb), d_b(vb), d_u(vb), d_r(vb), d_rh(vb), d_pv(vb), d_v(vb), d_s(vb), d_t(vb), d_tmp(sizeof(double) * 2 * static_cast<size_t>(n));
        const int nwg = (n + 255) / 256, nwg_vec = vec_nwg(n);
        // small systems are bound by dependent launches: the scalar steps travel with the kernels that consume them (LazyScalars,
        // tm_kernels.h; three partial-sum buffers in rotation, two scalar blocks) -- 5 launches per iteration instead of 9
        const int npart = std::max(nwg, nwg_vec);
        const bool lazy = npart <= 512 && !(opt.flags & TM_OPT_EAGER_SCALARS);
        Dev d_part(sizeof(double) * MAX_PARTIALS * static_cast<size_t>(npart) * (lazy ? 3 : 1)), d_red(sizeof(double) * MAX_PARTIALS), d_S(sizeof(KrylovScalars) * 2);
        hipStream_t st = nullptr;
        HIPCHK(hipMemcpyAsync(d_p.p, Ap, sizeof(int32_t) * (static_cast<size_t>(n) + 1), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_i.p, Ai, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_vx.p, Ax_x, sizeof(double) * nnz, hipMemcpyHostToDevice, st));
        if (two) HIPCHK(hipMemcpyAsync(d_vy.p, Ax_y, sizeof(double) * nnz, hipMemcpyHostToDevice, st));
        double* tmp = d_tmp.as<double>();
        auto upload2 = [&](const double* a, const double* b, double2* out) {   // two host arrays -> one interleaved device vector
            HIPCHK(hipMemcpyAsync(tmp, a, sizeof(double) * n, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(tmp + n, b, sizeof(double) * n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_interleave, dim3(nwg), dim3(256), 0, st, n, tmp, tmp + n, out);
            HIPCHK(hipGetLastError());
        };
        upload2(bx, by, d_b.as<double2>());
        upload2(x, y, d_u.as<double2>());   // warm start: the caller's x_new / y_new (BiCGStab.zig:136-153 seeds them from the coordinates)
        CsrDev A{n, d_p.as<int32_t>(), d_i.as<int32_t>(), d_vx.as<double>(), two ? d_vy.as<double>() : d_vx.as<double>(), d_dinv.as<double2>()};
        hipLaunchKernelGGL(k_csr_dinv, dim3(nwg), dim3(256), 0, st, n, A.p, A.i, A.vx, A.vy, d_dinv.as<double2>());
        HIPCHK(hipGetLastError());

        double* part_buf[3] = {d_part.as<double>(), d_part.as<double>() + (lazy ? 1 : 0) * MAX_PARTIALS * static_cast<size_t>(npart),
                               d_part.as<double>() + (lazy ? 2 : 0) * MAX_PARTIALS * static_cast<size_t>(npart)};
        int part_rot = 0;
        double* partials = part_buf[0];
        double* red = d_red.as<double>();
        KrylovScalars* S_buf[2] = {d_S.as<KrylovScalars>(), d_S.as<KrylovScalars>() + 1};
        KrylovScalars* S = S_buf[0];
        HIPCHK(hipMemsetAsync(S_buf[0], 0, sizeof(KrylovScalars) * 2, st));
        LazyStep pending[2];
        int npending = 0;
        auto flush_pending = [&]() {   // pending steps applied by launches of their own (before the host reads the scalars)
            for (int q = 0; q < npending; ++q) HIPCHK(launch_finalize_scalar(pending[q].partials, pending[q].nwg, red, S, pending[q].step, st));
            npending = 0;
        };
        auto reduce_update = [&](int nrows, int step) {   // the partial rows just written feed scalar step `step`
            if (!lazy) {
                HIPCHK(launch_finalize_scalar(partials, nrows, red, S, step, st));
                return;
            }
            if (npending == 2) flush_pending();
            pending[npending++] = LazyStep{step, partials, nrows};
            part_rot = (part_rot + 1) % 3;
            partials = part_buf[part_rot];   // the next producer writes elsewhere: this buffer is read by the consumer's workgroups
        };
        auto scalars_for = [&]() {   // for a kernel that reads the scalars: it applies the pending steps itself and publishes the result
            LazyScalars L;
            L.S_in = S;
            if (npending == 0) return L;
            KrylovScalars* other = (S == S_buf[0]) ? S_buf[1] : S_buf[0];
            L.S_out = other;
            L.nsteps = npending;
            for (int q = 0; q < npending; ++q) L.st[q] = pending[q];
            npending = 0;
            S = other;
            return L;
        };
        double2 *u = d_u.as<double2>(), *r = d_r.as<double2>(), *r_hat = d_rh.as<double2>(), *p = d_pv.as<double2>(), *v = d_v.as<double2>(),
                *s = d_s.as<double2>(), *t = d_t.as<double2>();
        hipLaunchKernelGGL(k_csr_bnorm, dim3(nwg), dim3(256), 0, st, n, d_b.as<double2>(), A.dinv, partials);
        HIPCHK(hipGetLastError());
        HIPCHK(launch_finalize_scalar(partials, nwg, red, S, STEP_TOL, st, opt.rtol, opt.atol));

        KrylovScalars h_S;
        auto read_S = [&]() {
            flush_pending();
            HIPCHK(hipMemcpyAsync(&h_S, S, sizeof(KrylovScalars), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        };
        uint64_t it_total = 0;
        int restarts = 0;
        bool converged = false, stalled = false;
        double rr0[2] = {0.0, 0.0};
        // stagnation watch, as in Smoother::poll_done: a recurrence residual that fails to halve over max(4000, 4 sqrt(n)) iterations ends
        // the solve (not converged, a warning) instead of running to the iteration cap
        double stall_best[2] = {0.0, 0.0};
        uint64_t stall_since = 0;
        const uint64_t stall_window = std::max<uint64_t>(4000, static_cast<uint64_t>(4.0 * std::sqrt(static_cast<double>(n))));
        while (true) {
            hipLaunchKernelGGL((k_csr_apply<true, DOT_OUT2>), dim3(nwg), dim3(256), 0, st, A, u, d_b.as<double2>(), nullptr, r, partials);
            HIPCHK(hipGetLastError());
            flush_pending();
            HIPCHK(launch_finalize_scalar(partials, nwg, red, S, STEP_INIT, st));
            HIPCHK(hipMemcpyAsync(r_hat, r, vb, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemsetAsync(p, 0, vb, st));
            HIPCHK(hipMemsetAsync(v, 0, vb, st));
            if (restarts == 0) {
                read_S();
                rr0[0] = h_S.rr0[0];
                rr0[1] = h_S.rr0[1];
                if (h_S.done[0] == 1 && h_S.done[1] == 1) {
                    converged = true;
                    break;
                }
            }
            bool breakdown = false;
            while (it_total < opt.max_inner) {
                HIPCHK(launch_p_update(scalars_for(), r, p, v, n, st));
                hipLaunchKernelGGL((k_csr_apply<false, DOT_AUX>), dim3(nwg), dim3(256), 0, st, A, p, nullptr, r_hat, v, partials);
                HIPCHK(hipGetLastError());
                reduce_update(nwg, STEP_SIGMA);
                HIPCHK(launch_s_update(scalars_for(), r, v, s, n, partials, st));
                reduce_update(nwg_vec, STEP_SS);
                hipLaunchKernelGGL((k_csr_apply<false, DOT_IN>), dim3(nwg), dim3(256), 0, st, A, s, nullptr, nullptr, t, partials);
                HIPCHK(hipGetLastError());
                reduce_update(nwg, STEP_TSTT);
                HIPCHK(launch_xr_update(scalars_for(), u, p, s, s, t, r, r_hat, n, partials, st));
                reduce_update(nwg_vec, STEP_RHO);
                it_total += 1;
                if (it_total % opt.check_every == 0 || it_total == opt.max_inner) {
                    read_S();
                    if (h_S.done[0] && h_S.done[1]) {
                        converged = h_S.done[0] == 1 && h_S.done[1] == 1;
                        breakdown = !converged;
                        break;
                    }
                    bool progress = false;
                    for (int c = 0; c < 2; ++c)
                        if (!h_S.done[c] && (!(stall_best[c] > 0.0) || h_S.rr[c] < 0.25 * stall_best[c])) {
                            stall_best[c] = h_S.rr[c];
                            progress = true;
                        }
                    if (progress) stall_since = it_total;
                    else if (it_total - stall_since > stall_window) {
                        stalled = true;
                        break;
                    }
                }
            }
            if (converged || !breakdown || stalled || restarts >= 8 || it_total >= opt.max_inner) break;
            restarts += 1;   // rho or omega vanished: restart from the current iterate (the reference only warns, BiCGStab.zig:368-369)
        }
        hipLaunchKernelGGL(k_deinterleave, dim3(nwg), dim3(256), 0, st, n, u, tmp, tmp + n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(x, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(y, tmp + n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            stats->outer_iterations = 1;
            stats->inner_iterations = it_total;
            stats->operator_sweeps = 1 + static_cast<uint64_t>(restarts) + 2 * it_total;
            stats->scaled_residual_rms = std::sqrt((rr0[0] + rr0[1]) / (2.0 * n));
            stats->not_converged = converged ? 0 : 1;
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        return converged ? TM_OK : TM_W_NOT_CONVERGED;
    });
}
