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
#include <array>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

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
                                                  const double* __restrict__ vy, double2* __restrict__ dinv, double2* __restrict__ dvec) {
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
    if (dvec) dvec[row] = make_double2(dx == 0.0 ? 1.0 : dx, dy == 0.0 ? 1.0 : dy);
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
        } else if (DOT == DOT_AUX2) {   // the operator acted on a preconditioned vector: t.s and t.t with the UNpreconditioned s in `aux`
            const double2 a = aux[row];
            acc[0] = a.x * o.x;
            acc[1] = a.y * o.y;
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

// partials of ||v||^2 per component
__global__ __launch_bounds__(256) void k_csr_norm2(int n, const double2* __restrict__ v, double* partials) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double acc[MAX_PARTIALS] = {0.0, 0.0, 0.0, 0.0};
    if (row < n) {
        const double2 w = v[row];
        acc[0] = w.x * w.x;
        acc[1] = w.y * w.y;
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

// ------------------------------------------------------------------------------------------------------------------
// ILU(0) on the device (TM_OPT_PRECOND_ILU0, seam 2 only): the reference's second preconditioner (preconditioner.zig:1-4; factorisation
// BiCGStab.zig:178-277, application :384-422 -- identical copies in GMRES.zig:199-298, 437-475), which its example inputs select.
// The recurrence is sequential row by row; what is parallel are the LEVEL SETS of its dependency graph: row i needs the finished rows
// k < i of its own pattern (factorisation and forward substitution alike), the backward substitution the rows k > i.  The host sorts the
// rows by level once per call (the pattern is the caller's); every row keeps the reference's operation order -- entries in CSR order,
// the update `lu[pos] -= l_ik * lu[k, j]` un-fused -- so the factor and M^-1 r equal the reference's bit for bit (tests/test_gpu_ilu0.py
// against the faithful oracle).  Levels of the reference's systems are narrow (T106: 1790 levels of ~14 rows): runs of levels with at most
// 256 rows execute inside ONE single-workgroup launch with a barrier between levels; a wider level gets a launch of its own.
// It is here so that the solver slot is complete and the reference's preconditioner can be cross-checked bit for bit, not for speed: a level
// is a chain of dependent loads (~2.6 us), T106's 2655 levels make an application ~7 ms and the solve 2.8 s where the diagonal-only solve
// takes 26 ms (tools/dev/csr_seam_probe.py; the CPU oracle: 65 ms per component).  Tried and dropped: fetching a level's matrix data one
// level ahead into per-thread arrays (the dynamic indexing sends them to scratch: 43 ms per iteration instead of 14).
struct IluDev {
    int n;
    const int32_t *p, *ci, *diag_pos;
    double *lux, *luy;       // the factors, in the pattern of A (unit lower part below the diagonal, U on and above it); luy == lux: one system
    const double2* dvec;     // a_ii per row and component: the forward substitution multiplies its right-hand side by it (see ilu_apply)
};
enum { ILU_FACTOR = 0, ILU_FORWARD = 1, ILU_BACKWARD = 2 };

__device__ __forceinline__ double ilu_pivot(const double* lu, int pos) {   // a missing or zero diagonal counts as 1 (BiCGStab.zig:240-249, 413-418)
    if (pos < 0) return 1.0;
    const double d = lu[pos];
    return d == 0.0 ? 1.0 : d;
}
__device__ __forceinline__ void ilu_factor_row(const IluDev& M, double* lu, int row) {
    const int start = M.p[row], end = M.p[row + 1];
    for (int idx = start; idx < end; ++idx) {
        const int col = M.ci[idx];
        if (col >= row) continue;
        const double lij = lu[idx] / ilu_pivot(lu, M.diag_pos[col]);
        lu[idx] = lij;
        for (int r = M.p[col]; r < M.p[col + 1]; ++r) {
            const int col_k = M.ci[r];
            if (col_k <= col) continue;
            for (int q = start; q < end; ++q)   // the reference's marker[]: where column col_k sits in THIS row, if it does
                if (M.ci[q] == col_k) {
                    lu[q] -= lij * lu[r];
                    break;
                }
        }
    }
}
template <int OP>
__device__ __forceinline__ void ilu_row(const IluDev& M, int row, const double2* __restrict__ rhs, double2* out) {
    if (OP == ILU_FACTOR) {
        ilu_factor_row(M, M.lux, row);
        if (M.luy != M.lux) ilu_factor_row(M, M.luy, row);
        return;
    }
    const int start = M.p[row], end = M.p[row + 1];
    if (OP == ILU_FORWARD) {
        double2 sum = rhs[row];
        if (M.dvec) {
            const double2 d = M.dvec[row];
            sum.x *= d.x;
            sum.y *= d.y;
        }
        for (int idx = start; idx < end; ++idx) {
            const int col = M.ci[idx];
            if (col < row) {
                const double2 o = out[col];
                sum.x -= M.lux[idx] * o.x;
                sum.y -= M.luy[idx] * o.y;
            }
        }
        out[row] = sum;
    } else {
        double2 sum = out[row];
        for (int idx = start; idx < end; ++idx) {
            const int col = M.ci[idx];
            if (col > row) {
                const double2 o = out[col];
                sum.x -= M.lux[idx] * o.x;
                sum.y -= M.luy[idx] * o.y;
            }
        }
        out[row] = make_double2(sum.x / ilu_pivot(M.lux, M.diag_pos[row]), sum.y / ilu_pivot(M.luy, M.diag_pos[row]));
    }
}
// levels [lv0, lv1) of `order` / `lev_ptr`.  One workgroup: every level has <= 256 rows, a barrier separates them (what a level reads
// was written by this workgroup).  Several workgroups: lv1 == lv0 + 1, one thread per row of that level.
template <int OP>
__global__ __launch_bounds__(256) void k_ilu_levels(IluDev M, const int32_t* __restrict__ order, const int32_t* __restrict__ lev_ptr, int lv0, int lv1,
                                                    const double2* __restrict__ rhs, double2* out) {
    if (gridDim.x > 1) {
        const int k = lev_ptr[lv0] + static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x);
        if (k < lev_ptr[lv0 + 1]) ilu_row<OP>(M, order[k], rhs, out);
        return;
    }
    for (int l = lv0; l < lv1; ++l) {
        const int k = lev_ptr[l] + static_cast<int>(threadIdx.x);
        if (k < lev_ptr[l + 1]) ilu_row<OP>(M, order[k], rhs, out);
        __syncthreads();   // (a workgroup-scope release / acquire: the next level reads what this one stored)
    }
}

// host side: level sets of the lower (factorisation, forward substitution) and of the upper (backward substitution) dependency graph
struct IluLevels {
    std::vector<int32_t> order, ptr;              // rows sorted by level; ptr[l] .. ptr[l+1]
    std::vector<std::array<int, 2>> chunks;       // launches: [lv0, lv1); a chunk of several levels has only levels of <= 256 rows
    void build(int n, const int32_t* p, const int32_t* ci, bool lower) {
        std::vector<int32_t> lev(static_cast<size_t>(n), 0);
        int32_t nlev = 0;
        auto visit = [&](int row) {
            int32_t l = 0;
            for (int k = p[row]; k < p[row + 1]; ++k) {
                const int col = ci[k];
                if (lower ? col < row : col > row) l = std::max(l, lev[col] + 1);
            }
            lev[row] = l;
            nlev = std::max(nlev, l + 1);
        };
        if (lower) for (int row = 0; row < n; ++row) visit(row);
        else for (int row = n - 1; row >= 0; --row) visit(row);
        ptr.assign(static_cast<size_t>(nlev) + 1, 0);
        for (int row = 0; row < n; ++row) ptr[lev[row] + 1] += 1;
        for (int l = 0; l < nlev; ++l) ptr[l + 1] += ptr[l];
        order.resize(static_cast<size_t>(n));
        std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
        for (int row = 0; row < n; ++row) order[at[lev[row]]++] = row;   // ascending row id inside a level
        chunks.clear();
        for (int l = 0; l < nlev;) {
            if (ptr[l + 1] - ptr[l] > 256) {
                chunks.push_back({l, l + 1});
                ++l;
                continue;
            }
            int e = l;
            while (e < nlev && ptr[e + 1] - ptr[e] <= 256 && e - l < (1 << 20)) ++e;
            chunks.push_back({l, e});
            l = e;
        }
    }
};

struct Dev {   // RAII device buffer
    void* p = nullptr;
    explicit Dev(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed (" + std::to_string(bytes) + " bytes)");
    }
    ~Dev() { (void)hipFree(p); }
    void reset(size_t bytes) {
        (void)hipFree(p);
        p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed (" + std::to_string(bytes) + " bytes)");
    }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

// ILU(0) of a system on the device: analysis on the host, factorisation and the two substitutions level by level (see k_ilu_levels)
struct IluState {
    IluLevels L, U;
    Dev d_diag, d_lux, d_luy, d_ordL, d_ptrL, d_ordU, d_ptrU;
    IluDev M{};
    IluState(int n, const int32_t* Ap, const int32_t* Ai, size_t nnz, bool two)
        : d_diag(sizeof(int32_t) * static_cast<size_t>(n)), d_lux(sizeof(double) * nnz), d_luy(two ? sizeof(double) * nnz : 0),
          d_ordL(sizeof(int32_t) * static_cast<size_t>(n)), d_ptrL(0), d_ordU(sizeof(int32_t) * static_cast<size_t>(n)), d_ptrU(0) {
        std::vector<int32_t> diag(static_cast<size_t>(n), -1);
        for (int row = 0; row < n; ++row)
            for (int k = Ap[row]; k < Ap[row + 1]; ++k)
                if (Ai[k] == row) {
                    diag[row] = k;
                    break;
                }
        L.build(n, Ap, Ai, true);
        U.build(n, Ap, Ai, false);
        HIPCHK(hipMemcpy(d_diag.p, diag.data(), sizeof(int32_t) * diag.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ordL.p, L.order.data(), sizeof(int32_t) * L.order.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ordU.p, U.order.data(), sizeof(int32_t) * U.order.size(), hipMemcpyHostToDevice));
        d_ptrL.reset(sizeof(int32_t) * L.ptr.size());
        d_ptrU.reset(sizeof(int32_t) * U.ptr.size());
        HIPCHK(hipMemcpy(d_ptrL.p, L.ptr.data(), sizeof(int32_t) * L.ptr.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ptrU.p, U.ptr.data(), sizeof(int32_t) * U.ptr.size(), hipMemcpyHostToDevice));
    }
    template <int OP>
    void run(const IluLevels& lv, const int32_t* order, const int32_t* ptr, const double2* rhs, double2* out, hipStream_t st) {
        for (const auto& c : lv.chunks) {
            const int width = lv.ptr[c[0] + 1] - lv.ptr[c[0]];
            const int grid = (c[1] - c[0] == 1 && width > 256) ? (width + 255) / 256 : 1;
            hipLaunchKernelGGL((k_ilu_levels<OP>), dim3(grid), dim3(256), 0, st, M, order, ptr, c[0], c[1], rhs, out);
            HIPCHK(hipGetLastError());
        }
    }
    // the factor of (vx, vy) -- device arrays in A's pattern -- into lux / luy
    void factor(int n, const int32_t* d_p, const int32_t* d_i, const double* d_vx, const double* d_vy, size_t nnz, const double2* dvec, hipStream_t st) {
        const bool two = d_vy != nullptr && d_vy != d_vx;
        HIPCHK(hipMemcpyAsync(d_lux.p, d_vx, sizeof(double) * nnz, hipMemcpyDeviceToDevice, st));
        if (two) HIPCHK(hipMemcpyAsync(d_luy.p, d_vy, sizeof(double) * nnz, hipMemcpyDeviceToDevice, st));
        M = IluDev{n, d_p, d_i, d_diag.as<int32_t>(), d_lux.as<double>(), two ? d_luy.as<double>() : d_lux.as<double>(), dvec};
        run<ILU_FACTOR>(L, d_ordL.as<int32_t>(), d_ptrL.as<int32_t>(), nullptr, nullptr, st);
    }
    // out = U^-1 L^-1 (dvec .* rhs)   (dvec == nullptr in M: plain M^-1 rhs, BiCGStab.zig:384-422)
    // times_d = false: plain M^-1 rhs whatever dvec the state was factorised with.  rhs == out is fine (a row reads its own right-hand side
    // before it stores, and nothing else of rhs)
    void apply(const double2* rhs, double2* out, hipStream_t st, bool times_d = true) {
        const double2* keep = M.dvec;
        if (!times_d) M.dvec = nullptr;
        run<ILU_FORWARD>(L, d_ordL.as<int32_t>(), d_ptrL.as<int32_t>(), rhs, out, st);
        run<ILU_BACKWARD>(U, d_ordU.as<int32_t>(), d_ptrU.as<int32_t>(), rhs, out, st);
        M.dvec = keep;
    }
};

}  // namespace

}  // namespace tmh

using namespace tmh;

// diagnostic (include/tm_hip_diag.h): ILU(0) of one CSR matrix on the device -- the factor in A's pattern and M^-1 rhs -- for the bit-for-bit
// comparison with the reference's recurrence (tests/test_gpu_ilu0.py)
extern "C" int tm_csr_ilu0_probe(uint64_t n64, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* rhs, double* lu_out, double* z_out) {
    return guarded([&]() {
        if (!Ap || !Ai || !Ax || !lu_out) throw TmError(TM_E_ARG, "null argument");
        if (n64 == 0 || n64 >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "system size out of range");
        {
            int dev = 0;
            HIPCHK(hipGetDevice(&dev));
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, dev));
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) throw TmError(TM_E_HIP, std::string("libtm_hip is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);
        }
        const int n = static_cast<int>(n64);
        const size_t nnz = static_cast<size_t>(Ap[n]);
        Dev d_p(sizeof(int32_t) * (static_cast<size_t>(n) + 1)), d_i(sizeof(int32_t) * nnz), d_v(sizeof(double) * nnz);
        HIPCHK(hipMemcpy(d_p.p, Ap, sizeof(int32_t) * (static_cast<size_t>(n) + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_i.p, Ai, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_v.p, Ax, sizeof(double) * nnz, hipMemcpyHostToDevice));
        IluState ilu(n, Ap, Ai, nnz, false);
        ilu.factor(n, d_p.as<int32_t>(), d_i.as<int32_t>(), d_v.as<double>(), nullptr, nnz, nullptr, nullptr);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(lu_out, ilu.d_lux.p, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        if (rhs && z_out) {
            std::vector<double2> r2(static_cast<size_t>(n));
            for (int k = 0; k < n; ++k) r2[k] = make_double2(rhs[k], rhs[k]);
            Dev d_r(sizeof(double2) * static_cast<size_t>(n)), d_z(sizeof(double2) * static_cast<size_t>(n));
            HIPCHK(hipMemcpy(d_r.p, r2.data(), sizeof(double2) * r2.size(), hipMemcpyHostToDevice));
            ilu.apply(d_r.as<double2>(), d_z.as<double2>(), nullptr);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(r2.data(), d_z.p, sizeof(double2) * r2.size(), hipMemcpyDeviceToHost));
            for (int k = 0; k < n; ++k) z_out[k] = r2[k].x;
        }
        return TM_OK;
    });
}

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
        if (opt.check_every == 0) opt.check_every = (opt.flags & TM_OPT_PRECOND_ILU0) ? 1 : 8;   // an ILU(0) iteration is thousands of dependent steps: poll each

        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) throw TmError(TM_E_HIP, std::string("libtm_hip is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);

        const bool two = Ax_y != nullptr && Ax_y != Ax_x;
        const bool use_ilu = (opt.flags & TM_OPT_PRECOND_ILU0) != 0;
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
        Dev d_dvec(use_ilu ? vb : 0), d_ph(use_ilu ? vb : 0), d_sh(use_ilu ? vb : 0);
        hipLaunchKernelGGL(k_csr_dinv, dim3(nwg), dim3(256), 0, st, n, A.p, A.i, A.vx, A.vy, d_dinv.as<double2>(), use_ilu ? d_dvec.as<double2>() : nullptr);
        HIPCHK(hipGetLastError());
        // ILU(0) as RIGHT preconditioner (BiCGStab.zig:314-316, 340-342) of the row-equilibrated operator B = D^-1 A the device iterates on:
        // B P^-1 ~ I with P = D^-1 M, M = ILU(0)(A) factorised from the caller's values exactly as the reference does, so P^-1 v = M^-1 (D v)
        // -- the forward substitution multiplies its right-hand side by the diagonal.  The stop test stays the scale-aware one.
        std::unique_ptr<IluState> ilu;
        if (use_ilu) {
            ilu = std::make_unique<IluState>(n, Ap, Ai, nnz, two);
            ilu->factor(n, A.p, A.i, A.vx, two ? A.vy : nullptr, nnz, d_dvec.as<double2>(), st);
        }

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

        if (opt.inner == TM_INNER_GMRES) {
            // The reference's other Krylov solver in the slot (GMRES.zig:300-423; csrc/tm_gmres.hip for the kernels): restarted GMRES(30), LEFT
            // preconditioned (GMRES.zig:335-336: z = M^-1 (A v)) with the diagonal -- the row-equilibrated operator the BiCGStab branch uses --
            // or with ILU(0): M^-1 A v = M^-1 (D (D^-1 A v)), the substitution multiplying its right-hand side by the diagonal.  Stop test:
            // GMRES's own residual norm |g_{j+1}| = ||P (b - A x)|| <= max(atol, rtol ||P b||), P the preconditioner (with the diagonal: the
            // scale-aware test of every other path).
            Dev d_W(vb), d_V(vb * (GMRES_M + 1)), d_G(sizeof(GmresScalars));
            double2 *W = d_W.as<double2>(), *V = d_V.as<double2>();
            GmresScalars* G = d_G.as<GmresScalars>();
            HIPCHK(hipMemsetAsync(G, 0, sizeof(GmresScalars), st));
            HIPCHK(hipMemsetAsync(V, 0, vb * (GMRES_M + 1), st));
            auto Vk = [&](int k) { return V + static_cast<size_t>(k) * n; };
            auto norm2_of = [&](const double2* v) {
                hipLaunchKernelGGL(k_csr_norm2, dim3(nwg), dim3(256), 0, st, n, v, partials);
                HIPCHK(hipGetLastError());
                HIPCHK(launch_finalize(partials, nwg, red, st));
            };
            if (use_ilu) {
                ilu->apply(d_b.as<double2>(), W, st, false);   // M^-1 b
                norm2_of(W);
            } else {
                hipLaunchKernelGGL(k_csr_bnorm, dim3(nwg), dim3(256), 0, st, n, d_b.as<double2>(), A.dinv, partials);
                HIPCHK(hipGetLastError());
                HIPCHK(launch_finalize(partials, nwg, red, st));
            }
            HIPCHK(launch_gm_tol(G, red, opt.rtol, opt.atol, st));
            GmresScalars h_G;
            auto poll = [&]() {
                HIPCHK(hipMemcpyAsync(&h_G, G, sizeof(GmresScalars), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                return h_G.done[0] == 1 && h_G.done[1] == 1;
            };
            uint64_t it_total = 0;
            bool converged = false, first = true;
            double rr0[2] = {0.0, 0.0};
            while (it_total < opt.max_inner) {
                hipLaunchKernelGGL((k_csr_apply<true, DOT_OUT2>), dim3(nwg), dim3(256), 0, st, A, u, d_b.as<double2>(), nullptr, W, partials);   // D^-1 (b - A x)
                HIPCHK(hipGetLastError());
                if (use_ilu) {
                    ilu->apply(W, W, st);   // M^-1 (b - A x)
                    norm2_of(W);
                } else {
                    HIPCHK(launch_finalize(partials, nwg, red, st));
                }
                HIPCHK(launch_gm_begin(G, red, st));
                const bool all_done = poll();
                if (first) {
                    rr0[0] = h_G.rr0[0];
                    rr0[1] = h_G.rr0[1];
                    first = false;
                }
                if (all_done) {
                    converged = true;
                    break;
                }
                HIPCHK(launch_gm_divide(Vk(0), W, G, n, st));
                bool cycle_done = false;
                for (int j = 0; j < GMRES_M && it_total < opt.max_inner; ++j) {
                    hipLaunchKernelGGL((k_csr_apply<false, DOT_NONE>), dim3(nwg), dim3(256), 0, st, A, Vk(j), nullptr, nullptr, W, partials);   // D^-1 A v_j
                    HIPCHK(hipGetLastError());
                    if (use_ilu) ilu->apply(W, W, st);
                    HIPCHK(launch_gm_mgs(W, nullptr, Vk(0), nullptr, G, 0, n, partials, st));
                    HIPCHK(launch_finalize(partials, nwg_vec, red, st));
                    for (int i = 1; i <= j; ++i) {
                        HIPCHK(launch_gm_mgs(W, Vk(i - 1), Vk(i), red, G, i - 1, n, partials, st));
                        HIPCHK(launch_finalize(partials, nwg_vec, red, st));
                    }
                    HIPCHK(launch_gm_mgs(W, Vk(j), nullptr, red, G, j, n, partials, st));
                    HIPCHK(launch_finalize(partials, nwg_vec, red, st));
                    HIPCHK(launch_gm_column(G, red, st));
                    HIPCHK(launch_gm_divide(Vk(j + 1), W, G, n, st));
                    it_total += 1;
                    if ((j + 1) % static_cast<int>(opt.check_every) == 0 || j + 1 == GMRES_M || it_total == opt.max_inner) {
                        if (poll()) {
                            cycle_done = true;
                            break;
                        }
                    }
                }
                HIPCHK(launch_gm_backsub(G, st));
                HIPCHK(launch_gm_update(u, V, n, G, n, st));
                if (cycle_done) {
                    converged = true;
                    break;
                }
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
                stats->operator_sweeps = it_total + (it_total + GMRES_M - 1) / GMRES_M + 1;
                stats->scaled_residual_rms = std::sqrt((rr0[0] + rr0[1]) / (2.0 * n));   // of P (b - A x0), P the preconditioner
                stats->not_converged = converged ? 0 : 1;
                stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            return converged ? TM_OK : TM_W_NOT_CONVERGED;
        }

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
                if (use_ilu) {   // the preconditioned recurrence, as Smoother::picard_bicgstab with the multigrid cycle in this place
                    double2 *p_hat = d_ph.as<double2>(), *s_hat = d_sh.as<double2>();
                    HIPCHK(launch_p_update(scalars_for(), r, p, v, n, st));
                    ilu->apply(p, p_hat, st);
                    hipLaunchKernelGGL((k_csr_apply<false, DOT_AUX>), dim3(nwg), dim3(256), 0, st, A, p_hat, nullptr, r_hat, v, partials);
                    HIPCHK(hipGetLastError());
                    reduce_update(nwg, STEP_SIGMA);
                    HIPCHK(launch_s_update(scalars_for(), r, v, s, n, partials, st));
                    reduce_update(nwg_vec, STEP_SS);
                    ilu->apply(s, s_hat, st);
                    hipLaunchKernelGGL((k_csr_apply<false, DOT_AUX2>), dim3(nwg), dim3(256), 0, st, A, s_hat, nullptr, s, t, partials);   // t.s, t.t with the unpreconditioned s
                    HIPCHK(hipGetLastError());
                    reduce_update(nwg, STEP_TSTT);
                    HIPCHK(launch_xr_update(scalars_for(), u, p_hat, s_hat, s, t, r, r_hat, n, partials, st));
                    reduce_update(nwg_vec, STEP_RHO);
                    it_total += 1;
                    if (it_total % opt.check_every == 0 || it_total == opt.max_inner) {
                        read_S();
                        if (h_S.done[0] && h_S.done[1]) {
                            converged = h_S.done[0] == 1 && h_S.done[1] == 1;
                            breakdown = !converged;
                            break;
                        }
                    }
                    continue;
                }
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
