// extern "C" surface of libtm_hip.so (declared in include/tm_hip.h; measurement / diagnostic entry points in include/tm_hip_diag.h).
#include "tm_smoother.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

using namespace tmh;

#include "tm_api_util.hpp"

static void require_gfx950() {
    static int checked = 0;
    if (checked) return;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        throw TmError(TM_E_HIP, std::string("libtm_hip is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);
    checked = 1;
}

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed");
    }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

static bool close2(const double* a, const double* b, double tol) { return std::fabs(a[0] - b[0]) <= tol && std::fabs(a[1] - b[1]) <= tol; }

template <class T>
static T* dup(const std::vector<T>& v) {
    T* p = static_cast<T*>(std::malloc(sizeof(T) * (v.empty() ? 1 : v.size())));
    if (!p) throw TmError(TM_E_MEMORY, "out of host memory");
    if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

extern "C" {

const char* tm_last_error(void) { return g_last_error.c_str(); }
int tm_abi_version(void) { return TM_HIP_ABI_VERSION; }
void tm_set_log(tm_log_fn sink, void* ctx) {
    g_log_sink = sink;
    g_log_ctx = ctx;
}

#ifdef TM_DEBUG_EXPORTS   // measurement build only (libtm_hip_dbg.so, tools/): not in include/tm_hip.h
// internal tuning knob used by the benchmark sweeps (not part of the drop-in surface)
int tm_tune_apply(int rows_per_chunk, int unroll, int pipe, int nt) {
    tune_apply(rows_per_chunk, unroll, pipe, nt);
    return TM_OK;
}

int tm_tune_fuse(int rows_per_chunk) {
    tune_fuse_rows(rows_per_chunk);
    return TM_OK;
}
// diagnostic: K2 with the same tiling / data movement but reduced arithmetic (mode 4 = copy, 5 = 9-point sum)
int tm_diag_apply(const double* d_in, double* d_out, uint64_t ni, uint64_t nj, int mode, void* stream) {
    return guarded([&]() {
        ApplyBlock a;
        a.in = reinterpret_cast<const double2*>(d_in);
        a.xk = a.in;
        a.pq = nullptr;
        a.aux = nullptr;
        a.out = reinterpret_cast<double2*>(d_out);
        a.ni = static_cast<int>(ni);
        a.nj = static_cast<int>(nj);
        a.omega = 1.0;
        a.partials = nullptr;
        HIPCHK(launch_apply_block(a, mode, DOT_NONE, static_cast<hipStream_t>(stream)));
        return TM_OK;
    });
}
#endif   // TM_DEBUG_EXPORTS

// ------------------------------------------------------------------ TFI (tfi.zig:112-208)
// diagnostic (include/tm_hip_diag.h): what this GPU sustains on a plain copy / triad of `bytes` per array, vector-kernel access pattern.
// Runs on a private non-blocking stream (the legacy null stream would synchronise with every blocking stream of the process, and a
// multi-rank job calls this on rank 0 while the other ranks move on); stream and events are released on every path.
namespace {
struct ProbeStream {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProbeStream() {
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
    }
    ~ProbeStream() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
    ProbeStream(const ProbeStream&) = delete;
    ProbeStream& operator=(const ProbeStream&) = delete;
};
}  // namespace
int tm_stream_probe(uint64_t bytes, int32_t iters, double* copy_GBps, double* triad_GBps) {
    return guarded([&]() {
        if (bytes < 4096 || iters < 1 || !copy_GBps || !triad_GBps) throw TmError(TM_E_ARG, "bad argument");
        require_gfx950();
        const int64_t n = static_cast<int64_t>(bytes / sizeof(double2));
        DevBuf a(sizeof(double2) * n), b(sizeof(double2) * n), c(sizeof(double2) * n);
        ProbeStream ps;
        HIPCHK(hipMemsetAsync(a.p, 0, sizeof(double2) * n, ps.s));
        HIPCHK(hipMemsetAsync(b.p, 0, sizeof(double2) * n, ps.s));
        HIPCHK(hipMemsetAsync(c.p, 0, sizeof(double2) * n, ps.s));
        double out[2] = {0.0, 0.0};
        for (int kind = 0; kind < 2; ++kind) {
            // the direction alternates (a <- b, b <- a, ...), like the two fields of a relaxation sweep: a source that no launch ever writes
            // would be served from the 256 MB Infinity Cache and read as "8.5 TB/s" (tools/ubench/stream.hip)
            auto once = [&](int k) {
                double2* v[3] = {a.as<double2>(), b.as<double2>(), c.as<double2>()};
                if (kind == 0) return launch_stream(0, v[k & 1], v[(k & 1) ^ 1], v[2], 0.5, n, ps.s);
                return launch_stream(1, v[k % 3], v[(k + 1) % 3], v[(k + 2) % 3], 0.5, n, ps.s);   // the three arrays rotate
            };
            for (int w = 0; w < 4; ++w) HIPCHK(once(w));
            HIPCHK(hipEventRecord(ps.e0, ps.s));
            for (int k = 0; k < iters; ++k) HIPCHK(once(k));
            HIPCHK(hipEventRecord(ps.e1, ps.s));
            HIPCHK(hipEventSynchronize(ps.e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ps.e0, ps.e1));
            out[kind] = static_cast<double>(kind == 1 ? 3 : 2) * sizeof(double2) * n * iters / (1e-3 * ms) / 1e9;
        }
        *copy_GBps = out[0];
        *triad_GBps = out[1];
        return TM_OK;
    });
}

// diagnostic (include/tm_hip_diag.h, tests/test_gpu_refmath.py): acos(x[i]) and atan2(y[i], x[i]) as the device's White kernels evaluate
// them (tm_refmath.h), host arrays in and out
int tm_white_math_probe(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2) {
    return guarded([&]() {
        if (!x || !y || !out_acos || !out_atan2) throw TmError(TM_E_ARG, "null argument");
        require_gfx950();
        const size_t nb = sizeof(double) * n;
        DevBuf dx(nb), dy(nb), da(nb), dt(nb);
        HIPCHK(hipMemcpy(dx.p, x, nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dy.p, y, nb, hipMemcpyHostToDevice));
        HIPCHK(launch_debug_white_math(dx.as<double>(), dy.as<double>(), n, da.as<double>(), dt.as<double>(), nullptr));
        HIPCHK(hipMemcpy(out_acos, da.p, nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out_atan2, dt.p, nb, hipMemcpyDeviceToHost));
        return TM_OK;
    });
}

int tm_tfi_block(double* xy_out, uint64_t ni, uint64_t nj, const double* x_i_min, const double* x_i_max, const double* x_j_min,
                 const double* x_j_max, const double* s1, const double* s2, const double* t1, const double* t2) {
    return guarded([&]() {
        if (!xy_out || !x_i_min || !x_i_max || !x_j_min || !x_j_max || !s1 || !s2 || !t1 || !t2) throw TmError(TM_E_ARG, "null argument");
        if (ni < 2 || nj < 2 || ni * nj >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "InconsistentSize: block must be at least 2 x 2 and below 2^31 nodes");
        // the reference's debug asserts (tfi.zig:135-162) become error codes
        if (s1[0] != 0 || s1[ni - 1] != 1.0 || s2[0] != 0 || s2[ni - 1] != 1.0 || t1[0] != 0 || t1[nj - 1] != 1.0 || t2[0] != 0 ||
            t2[nj - 1] != 1.0)
            throw TmError(TM_E_ARG, "clusterings must start at 0 and end at exactly 1 (tfi.zig:135-145)");
        const double tol = 1e-10;
        if (!close2(x_i_min, x_j_min, tol) || !close2(x_i_min + 2 * (ni - 1), x_j_max, tol) || !close2(x_j_min + 2 * (nj - 1), x_i_max, tol) ||
            !close2(x_i_max + 2 * (ni - 1), x_j_max + 2 * (nj - 1), tol))
            throw TmError(TM_E_MISMATCH, "edge end points do not meet at the block corners (tfi.zig:150-162)");
        require_gfx950();
        const size_t ei = sizeof(double) * 2 * ni, ej = sizeof(double) * 2 * nj;
        DevBuf out(sizeof(double) * 2 * ni * nj), a(ei), b(ei), c(ej), d(ej), ds1(ei / 2), ds2(ei / 2), dt1(ej / 2), dt2(ej / 2);
        HIPCHK(hipMemcpy(a.p, x_i_min, ei, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b.p, x_i_max, ei, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c.p, x_j_min, ej, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.p, x_j_max, ej, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ds1.p, s1, ei / 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ds2.p, s2, ei / 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dt1.p, t1, ej / 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dt2.p, t2, ej / 2, hipMemcpyHostToDevice));
        HIPCHK(launch_tfi_block(out.as<double2>(), static_cast<int>(ni), static_cast<int>(nj), a.as<double2>(), b.as<double2>(), c.as<double2>(),
                                d.as<double2>(), ds1.as<double>(), ds2.as<double>(), dt1.as<double>(), dt2.as<double>(), nullptr));
        HIPCHK(hipMemcpy(xy_out, out.p, sizeof(double) * 2 * ni * nj, hipMemcpyDeviceToHost));
        return TM_OK;
    });
}

int tm_tfi_linear2d(double* xy_out, uint64_t ni, uint64_t nj, const double* e_i_min, const double* e_i_max, const double* e_j_min,
                    const double* e_j_max) {
    return guarded([&]() {
        if (!xy_out || !e_i_min || !e_i_max || !e_j_min || !e_j_max) throw TmError(TM_E_ARG, "null argument");
        if (ni < 2 || nj < 2 || ni * nj >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "InconsistentSize (tfi.zig:30)");
        require_gfx950();
        const size_t ei = sizeof(double) * 2 * ni, ej = sizeof(double) * 2 * nj;
        DevBuf out(sizeof(double) * 2 * ni * nj), a(ei), b(ei), c(ej), d(ej);
        HIPCHK(hipMemcpy(a.p, e_i_min, ei, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b.p, e_i_max, ei, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c.p, e_j_min, ej, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.p, e_j_max, ej, hipMemcpyHostToDevice));
        HIPCHK(launch_tfi_linear2d(out.as<double2>(), static_cast<int>(ni), static_cast<int>(nj), a.as<double2>(), b.as<double2>(), c.as<double2>(),
                                   d.as<double2>(), nullptr));
        HIPCHK(hipMemcpy(xy_out, out.p, sizeof(double) * 2 * ni * nj, hipMemcpyDeviceToHost));
        return TM_OK;
    });
}

int tm_dev_tfi_block(double* d_xy, uint64_t ni, uint64_t nj, const double* a, const double* b, const double* c, const double* d,
                     const double* s1, const double* s2, const double* t1, const double* t2, void* stream) {
    return guarded([&]() {
        if (ni < 2 || nj < 2 || ni * nj >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "InconsistentSize");
        HIPCHK(launch_tfi_block(reinterpret_cast<double2*>(d_xy), static_cast<int>(ni), static_cast<int>(nj), reinterpret_cast<const double2*>(a),
                                reinterpret_cast<const double2*>(b), reinterpret_cast<const double2*>(c), reinterpret_cast<const double2*>(d), s1, s2,
                                t1, t2, static_cast<hipStream_t>(stream)));
        return TM_OK;
    });
}

uint64_t tm_dev_relax_partials_needed(uint64_t ni, uint64_t nj) {
    if (ni < 3 || nj < 3) return 0;
    return static_cast<uint64_t>(apply_block_nwg(static_cast<int>(ni), static_cast<int>(nj))) * MAX_PARTIALS;
}

int tm_dev_relax_sweep(const double* d_in, double* d_out, uint64_t ni, uint64_t nj, double omega, double* d_partials,
                       uint64_t partials_capacity, uint64_t* nwg, void* stream) {
    return guarded([&]() {
        if (ni < 3 || nj < 3 || ni * nj >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "block must be at least 3 x 3");
        if (d_in == d_out) throw TmError(TM_E_ARG, "a Jacobi sweep cannot run in place");
        const uint64_t need = tm_dev_relax_partials_needed(ni, nj);
        if (d_partials && partials_capacity < need) throw TmError(TM_E_ARG, "partials buffer too small");
        ApplyBlock a;
        a.in = reinterpret_cast<const double2*>(d_in);
        a.xk = a.in;
        a.pq = nullptr;
        a.aux = nullptr;
        a.out = reinterpret_cast<double2*>(d_out);
        a.ni = static_cast<int>(ni);
        a.nj = static_cast<int>(nj);
        a.omega = omega;
        a.partials = d_partials;
        hipStream_t st = static_cast<hipStream_t>(stream);
        HIPCHK(launch_apply_block(a, MODE_RELAX, d_partials ? DOT_DELTA : DOT_NONE, st));
        HIPCHK(launch_copy_perimeter(a.in, a.out, a.ni, a.nj, st));
        if (nwg) *nwg = need / MAX_PARTIALS;
        return TM_OK;
    });
}

// ------------------------------------------------------------------ handle
int tm_smoother_create(const tm_mesh_desc* mesh, const tm_solver_opt* opt, const tm_control_fn* cf, const tm_comm_hooks* hooks, void* stream,
                       tm_smoother** out) {
    return guarded([&]() {
        if (!out) throw TmError(TM_E_ARG, "null output handle");
        *out = nullptr;
        require_gfx950();
        auto h = std::make_unique<tm_smoother>();
        h->impl.create(mesh, opt, cf, hooks, stream, false);
        *out = h.release();
        return TM_OK;
    });
}
int tm_smoother_workspace_bytes(const tm_mesh_desc* mesh, const tm_solver_opt* opt, const tm_control_fn* cf, const tm_comm_hooks* hooks,
                                uint64_t* bytes) {
    return guarded([&]() {
        if (!bytes) throw TmError(TM_E_ARG, "null output");
        tm_smoother tmp;
        tmp.impl.create(mesh, opt, cf, hooks, nullptr, true);
        *bytes = tmp.impl.arena.used() + 4096;
        return TM_OK;
    });
}
void tm_smoother_destroy(tm_smoother* s) {
    delete s;   // ~Smoother releases the pinned buffers, events, side stream and scratch
}
int tm_smoother_iterate(tm_smoother* s, uint64_t iterations, tm_stats* stats) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        tm_stats st;
        s->impl.iterate(iterations, &st);
        if (stats) *stats = st;
        return st.not_converged ? TM_W_NOT_CONVERGED : TM_OK;
    });
}
int tm_smoother_iterate_until(tm_smoother* s, uint64_t max_iterations, double scaled_residual_tol, tm_stats* stats) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        if (!(scaled_residual_tol > 0.0)) throw TmError(TM_E_ARG, "the residual tolerance must be positive");
        tm_stats st;
        const bool reached = s->impl.iterate_until(max_iterations, scaled_residual_tol, &st);
        if (stats) *stats = st;
        return (reached && !st.not_converged) ? TM_OK : TM_W_NOT_CONVERGED;
    });
}
int tm_smoother_iterate_until_update(tm_smoother* s, uint64_t max_iterations, double update_rms_tol, tm_stats* stats) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        if (!(update_rms_tol > 0.0)) throw TmError(TM_E_ARG, "the update tolerance must be positive");
        tm_stats st;
        const bool reached = s->impl.iterate_until_update(max_iterations, update_rms_tol, &st);
        if (stats) *stats = st;
        return (reached && !st.not_converged) ? TM_OK : TM_W_NOT_CONVERGED;
    });
}
int tm_smoother_download(tm_smoother* s, const tm_mesh_desc* mesh) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        s->impl.download(mesh);
        return TM_OK;
    });
}
int tm_smoother_upload(tm_smoother* s, const tm_mesh_desc* mesh) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        s->impl.upload(mesh);
        s->impl.outer_done = 0;
        if (s->impl.white) {
            HIPCHK(hipMemsetAsync(s->impl.PQ, 0, sizeof(double2) * s->impl.n_local, s->impl.stream));
            s->impl.white_launch(0);
            s->impl.sync();
        }
        return TM_OK;
    });
}
int tm_smoother_exchange_plan(const tm_smoother* s, int32_t* npeers, const int32_t** peer_rank, const int64_t** send_offset,
                              const int64_t** send_count, const int64_t** recv_offset, const int64_t** recv_count) {
    return guarded([&]() {
        if (!s || !npeers) throw TmError(TM_E_ARG, "null argument");
        const LocalPlan& lp = s->impl.lp;
        *npeers = static_cast<int32_t>(lp.peer_rank.size());
        if (peer_rank) *peer_rank = lp.peer_rank.data();
        if (send_offset) *send_offset = lp.direct_send ? lp.send_first.data() : lp.send_off.data();
        if (send_count) *send_count = lp.send_cnt.data();
        if (recv_offset) *recv_offset = lp.recv_off.data();
        if (recv_count) *recv_count = lp.recv_cnt.data();
        return TM_OK;
    });
}
int tm_smoother_apply(tm_smoother* s, const double* in_xy, double* out_xy, int scaled) {
    return guarded([&]() {
        if (!s || !in_xy || !out_xy) throw TmError(TM_E_ARG, "null argument");
        s->impl.apply_host(in_xy, out_xy, scaled);
        return TM_OK;
    });
}
int tm_smoother_assemble_csr(tm_smoother* s, int32_t* Ap, int32_t* Ai, double* Ax_x, double* Ax_y, uint64_t nnz_capacity, uint64_t* nnz) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        const uint64_t n = s->impl.assemble_csr_host(Ap, Ai, Ax_x, Ax_y, nnz_capacity);
        if (nnz) *nnz = n;
        return TM_OK;
    });
}
int tm_smoother_apply_reference_order(tm_smoother* s, const double* in_xy, double* out_xy) {
    return guarded([&]() {
        if (!s || !in_xy || !out_xy) throw TmError(TM_E_ARG, "null argument");
        s->impl.apply_reference_host(in_xy, out_xy);
        return TM_OK;
    });
}
int tm_smoother_rhs(tm_smoother* s, double* rhs_xy) {
    return guarded([&]() {
        if (!s || !rhs_xy) throw TmError(TM_E_ARG, "null argument");
        s->impl.rhs_host(rhs_xy);
        return TM_OK;
    });
}
int tm_smoother_row_kinds(const tm_smoother* s, int32_t* kinds) {
    return guarded([&]() {
        if (!s || !kinds) throw TmError(TM_E_ARG, "null argument");
        for (int64_t i = 0; i < s->impl.topo.dof; ++i) kinds[i] = -1;
        for (const PlanRow& r : s->impl.all_rows) kinds[r.gid] = r.kind;
        return TM_OK;
    });
}
uint64_t tm_smoother_dof(const tm_smoother* s) { return s ? static_cast<uint64_t>(s->impl.topo.dof) : 0; }
int tm_smoother_control_function(tm_smoother* s, double* pq) {
    return guarded([&]() {
        if (!s || !pq) throw TmError(TM_E_ARG, "null argument");
        s->impl.control_function_host(pq);
        return TM_OK;
    });
}

int tm_smoother_export_soa(tm_smoother* s, uint64_t block, double* x, double* y, double* p, double* q) {
    return guarded([&]() {
        if (!s || !x || !y || ((p == nullptr) != (q == nullptr))) throw TmError(TM_E_ARG, "null argument (p and q come together)");
        s->impl.export_soa_host(static_cast<int64_t>(block), x, y, p, q);
        return TM_OK;
    });
}

int tm_export_soa(const double* xy, uint64_t ni, uint64_t nj, double* x_out, double* y_out) {
    return guarded([&]() {
        if (!xy || !x_out || !y_out) throw TmError(TM_E_ARG, "null argument");
        if (ni < 1 || nj < 1 || ni * nj >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "block size out of range");
        require_gfx950();
        const size_t n = static_cast<size_t>(ni) * nj;
        DevBuf in(sizeof(double) * 2 * n), out(sizeof(double) * 2 * n);
        HIPCHK(hipMemcpy(in.p, xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
        HIPCHK(launch_soa_planes(in.as<double2>(), out.as<double>(), out.as<double>() + n, static_cast<int>(ni), static_cast<int>(nj), nullptr));
        HIPCHK(hipMemcpy(x_out, out.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(y_out, out.as<double>() + n, sizeof(double) * n, hipMemcpyDeviceToHost));
        return TM_OK;
    });
}

int tm_smoother_profile(tm_smoother* s, int enable) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        s->impl.profile = enable > 0 ? enable : 0;
        s->impl.ev_used = 0;
        s->impl.prof_launches = 0;
        s->impl.prof_timed = 0;
        s->impl.prof_phase = 0;
        s->impl.prof_open = 0;
        return TM_OK;
    });
}
int tm_smoother_profile_read(tm_smoother* s, double* k2_ms_total, uint64_t* k2_launches_timed, uint64_t* k2_launches) {
    return guarded([&]() {
        if (!s) throw TmError(TM_E_ARG, "null handle");
        s->impl.profile_read(k2_ms_total, k2_launches, k2_launches_timed);
        return TM_OK;
    });
}

int tm_smoother_inner(const tm_smoother* s) { return s ? s->impl.opt.inner : static_cast<int>(TM_E_ARG); }

// how the handle orders the two queues of a pipelined pass (include/tm_hip_diag.h)
int tm_smoother_queue_ordering(const tm_smoother* s) { return s ? s->impl.queue_ordering : static_cast<int>(TM_E_ARG); }

// ------------------------------------------------------------------ seam 1 (smooth.zig:74-80)
int tm_smooth_mesh(const tm_mesh_desc* mesh, uint64_t iterations, const tm_solver_opt* opt, const tm_control_fn* cf, tm_stats* stats) {
    tm_smoother* h = nullptr;
    int rc = tm_smoother_create(mesh, opt, cf, nullptr, nullptr, &h);
    if (rc < 0) return rc;
    tm_stats st;
    std::memset(&st, 0, sizeof(st));
    rc = tm_smoother_iterate(h, iterations, &st);
    if (rc >= 0 && iterations > 0) {   // iterations == 0 returns the mesh untouched (input.zig:28 default)
        const int rd = tm_smoother_download(h, mesh);
        if (rd < 0) rc = rd;
    }
    tm_smoother_destroy(h);
    if (stats) *stats = st;
    return rc;
}

// ------------------------------------------------------------------ host-only planning export
int tm_plan_local(const tm_mesh_desc* mesh, const int32_t* owner, int32_t rank, int32_t nranks, tm_plan_local_info* out) {
    return guarded([&]() {
        if (!mesh || !owner || !out) throw TmError(TM_E_ARG, "null argument");
        std::memset(out, 0, sizeof(*out));
        const Topology t = topo_of(mesh);
        const std::vector<PlanRow> rows = build_rows(t);
        const std::vector<int32_t> own(owner, owner + t.nblocks());
        const LocalPlan lp = build_local_plan(t, rows, own, rank, nranks);
        out->n_owned = lp.n_owned;
        out->n_ghost = static_cast<int64_t>(lp.ghost_gid.size());
        out->n_send = static_cast<int64_t>(lp.send_ids.size());
        out->npeers = static_cast<int32_t>(lp.peer_rank.size());
        out->nowned_blocks = static_cast<int32_t>(lp.owned_blocks.size());
        std::vector<int64_t> send_gid;
        for (int32_t l : lp.send_ids) {
            size_t k = lp.owned_blocks.size() - 1;
            while (l < lp.local_start[k]) --k;
            send_gid.push_back(t.start[lp.owned_blocks[k]] + (l - lp.local_start[k]));
        }
        out->owned_blocks = dup(lp.owned_blocks);
        out->local_start = dup(lp.local_start);
        out->ghost_gid = dup(lp.ghost_gid);
        out->send_ids = dup(lp.send_ids);
        out->send_gid = dup(send_gid);
        out->peer_rank = dup(lp.peer_rank);
        out->send_offset = dup(lp.send_off);
        out->send_count = dup(lp.send_cnt);
        out->recv_offset = dup(lp.recv_off);
        out->recv_count = dup(lp.recv_cnt);
        out->send_first = dup(lp.send_first);
        out->direct_send = lp.direct_send ? 1 : 0;
        std::vector<int64_t> gg, gc;
        std::vector<int32_t> gk;
        for (const PlanRow& g : lp.ghost_rows) {
            gg.push_back(g.gid);
            gk.push_back(g.kind);
            for (int q = 0; q < 9; ++q) gc.push_back(q < g.ncols ? g.col[q] : -1);
        }
        out->n_ghost_rows = static_cast<int64_t>(gg.size());
        out->ghost_row_gid = dup(gg);
        out->ghost_row_kind = dup(gk);
        out->ghost_row_cols = dup(gc);
        return TM_OK;
    });
}
void tm_plan_local_free(tm_plan_local_info* i) {
    if (!i) return;
    std::free(i->owned_blocks);
    std::free(i->local_start);
    std::free(i->ghost_gid);
    std::free(i->send_ids);
    std::free(i->send_gid);
    std::free(i->peer_rank);
    std::free(i->send_first);
    std::free(i->send_offset);
    std::free(i->send_count);
    std::free(i->recv_offset);
    std::free(i->recv_count);
    std::free(i->ghost_row_gid);
    std::free(i->ghost_row_kind);
    std::free(i->ghost_row_cols);
    std::memset(i, 0, sizeof(*i));
}

int tm_plan_build(const tm_mesh_desc* mesh, tm_plan_rows* out) {
    return guarded([&]() {
        if (!mesh || !out) throw TmError(TM_E_ARG, "null argument");
        std::memset(out, 0, sizeof(*out));
        const Topology t = topo_of(mesh);
        const std::vector<PlanRow> rows = build_rows(t);
        const size_t n = rows.size();
        out->nrows = n;
        out->row = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (n ? n : 1)));
        out->kind = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (n ? n : 1)));
        out->ncols = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (n ? n : 1)));
        out->cols = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * 9 * (n ? n : 1)));
        out->coef_x = static_cast<double*>(std::malloc(sizeof(double) * 9 * (n ? n : 1)));
        out->coef_y = static_cast<double*>(std::malloc(sizeof(double) * 9 * (n ? n : 1)));
        out->rhs = static_cast<double*>(std::malloc(sizeof(double) * 2 * (n ? n : 1)));
        out->slot = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * 9 * (n ? n : 1)));
        if (!out->row || !out->kind || !out->ncols || !out->cols || !out->coef_x || !out->coef_y || !out->rhs || !out->slot) {
            tm_plan_free(out);
            throw TmError(TM_E_MEMORY, "out of host memory");
        }
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (size_t k = 0; k < n; ++k) {
            const PlanRow& r = rows[k];
            out->row[k] = r.gid;
            out->kind[k] = r.kind;
            out->ncols[k] = r.ncols;
            for (int q = 0; q < 9; ++q) {
                out->cols[k * 9 + q] = q < r.ncols ? r.col[q] : -1;
                out->coef_x[k * 9 + q] = (r.kind == KIND_SMOOTHED) ? nan : r.cx[q];
                out->coef_y[k * 9 + q] = (r.kind == KIND_SMOOTHED) ? nan : r.cy[q];
                out->slot[k * 9 + q] = r.slot[q];
            }
            out->rhs[2 * k] = (r.rhs_coord & 1) ? nan : r.rhs[0];
            out->rhs[2 * k + 1] = (r.rhs_coord & 2) ? nan : r.rhs[1];
        }
        return TM_OK;
    });
}
void tm_plan_free(tm_plan_rows* rows) {
    if (!rows) return;
    std::free(rows->row);
    std::free(rows->kind);
    std::free(rows->ncols);
    std::free(rows->cols);
    std::free(rows->coef_x);
    std::free(rows->coef_y);
    std::free(rows->rhs);
    std::free(rows->slot);
    std::memset(rows, 0, sizeof(*rows));
}

}  // extern "C"
