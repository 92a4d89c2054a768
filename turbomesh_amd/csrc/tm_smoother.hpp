// Device-resident smoother handle (internal).  One handle = one rank's share of a mesh.
#pragma once
#include "../../include/tm_hip.h"
#include <cstdlib>
#include "tm_kernels.h"
#include "tm_multigrid.hpp"
#include "tm_plan.hpp"
#include <cmath>
#include <atomic>
#include <memory>
#include <string>
#include <functional>
#include <vector>

namespace tmh {

struct TmError : std::runtime_error {
    int code;
    TmError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
void hip_check(hipError_t e, const char* what);
// per-iteration log sink (tm_set_log), process-wide like the reference's std.log
extern std::atomic<int> g_multirank_handles;   // live handles with comm hooks in this process
extern tm_log_fn g_log_sink;
extern void* g_log_ctx;

// Bump allocator over either library-owned hipMalloc chunks or a caller-provided workspace.
class DeviceArena {
   public:
    DeviceArena() = default;
    ~DeviceArena();
    void use_workspace(void* base, uint64_t bytes);   // carve from caller memory
    void measure_only();                              // only count bytes (tm_smoother_workspace_bytes)
    void* alloc(uint64_t bytes);
    template <class T>
    T* alloc_n(uint64_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    uint64_t used() const { return used_; }
    bool measuring() const { return measure_; }

   private:
    std::vector<void*> owned_;
    char* base_ = nullptr;
    uint64_t cap_ = 0, used_ = 0;
    bool external_ = false, measure_ = false;
};

struct Smoother {
    Smoother() = default;
    ~Smoother();
    Smoother(const Smoother&) = delete;
    Smoother& operator=(const Smoother&) = delete;
    Topology topo;
    std::vector<PlanRow> all_rows;
    LocalPlan lp;
    tm_solver_opt opt{};
    tm_control_fn cf{};
    tm_comm_hooks hooks{};
    bool has_hooks = false;
    std::vector<int32_t> owner;
    hipStream_t stream = nullptr;
    DeviceArena arena;

    int64_t n_owned = 0, n_local = 0, n_ghost = 0, dof_global = 0;
    // vectors, double2[n_local]
    double2 *X = nullptr, *U = nullptr, *r = nullptr, *r_hat = nullptr, *p = nullptr, *v = nullptr, *s = nullptr, *t = nullptr;
    double2 *PQ = nullptr, *tmpA = nullptr, *tmpB = nullptr;
    // multigrid preconditioner (TM_INNER_MG_BICGSTAB): preconditioned search directions, two scratch vectors, one hierarchy per owned block
    double2 *p_hat = nullptr, *s_hat = nullptr, *mg_w0 = nullptr, *mg_w1 = nullptr;
    std::vector<BlockMG> mg;
    bool use_mg = false;
    bool mg_perimeter_step = false;   // the preconditioner applies the perimeter rows to the interior corrections (precondition())
    bool mg_dirichlet = false;   // ... and hands the perimeter values to the cycles as Dirichlet data (precondition())
    // the cycles of a multi-block mesh run side by side: one stream per block (at most 8), forked from and joined to the handle's stream by events
    std::vector<hipStream_t> mg_streams;
    std::vector<hipEvent_t> mg_join;
    hipEvent_t mg_fork = nullptr;
    int mg_perimeter_sweeps = 2;   // passes of the perimeter rows behind the cycles: the first on (e_I, 0), the others Jacobi sweeps on the perimeter system
    void precondition(const double2* in, double2* out);
    double2* M = nullptr;           // X^(k+1) of a fused pair of relax sweeps (perimeter + first-interior ring only)
    double2* M2 = nullptr;          // coupled triples: X^(k+2) on the perimeter and in the zone next to sides whose perimeter rows move
    bool fuse_pairs = false;
    bool relax2_store_nt = true;     // K2x2 result stores streaming (nt) or plain: by the rank's footprint against the Infinity Cache (create())
    bool pair_sync_events = false;   // TM_PAIR_SYNC=events when the handle was created: multi-rank sweep pairs ordered by events, not device counters
    size_t inside_lds = 0;          // dynamic LDS of the interior pass of a multi-rank sweep pair (occupancy cap, see create())
    // perimeter rows
    EdgeRowsDev edge;
    std::vector<double> h_rhs;   // host copy of the static rhs (refilled on upload), lp.rows order
    std::vector<int32_t> order_all, order_nf, order_nf_g;   // run order of the three row tables (see build_table)
    double* d_rhs = nullptr;
    // relax mode: the perimeter rows that are not `fixed` (the only ones a sweep has to evaluate), and per owned block which
    // sides carry such rows
    EdgeRowsDev edge_nf;
    double* d_rhs_nf = nullptr;
    EdgeRowsDev edge_nf_g;          // the same + the depth-1 ghost rows (multi-rank sweep pairs; empty otherwise)
    double* d_rhs_nf_g = nullptr;
    std::vector<size_t> nf_rows;
    std::vector<int> dyn_mask;
    std::vector<const int32_t*> border_ids;   // per owned block: device list of the K2x2 border tiles
    std::vector<int> border_n;
    void prefill_fixed();   // perimeter of X -> perimeter of U and M
    // reductions
    double* partials = nullptr;
    double* red = nullptr;
    KrylovScalars* S = nullptr;
    KrylovScalars* h_S = nullptr;   // pinned
    // pipelined convergence poll (poll_done): two pinned copies of the scalar block in flight alternately
    bool pipelined_poll = false, poll_open = false;
    KrylovScalars* h_poll[2] = {nullptr, nullptr};
    hipEvent_t ev_poll[2] = {nullptr, nullptr};
    uint64_t poll_it[2] = {0, 0}, poll_iters = 0;
    int poll_slot = 0;
    int poll_done(uint64_t it, bool final);
    double stall_best[2] = {0.0, 0.0};   // stagnation watch of the inner solve (poll_done): best ||r||^2 per component, iteration of the last halving
    uint64_t stall_since = 0, stall_window = 4000;
    bool stalled = false;
    double* h_red = nullptr;        // pinned
    // the two-kernel Krylov iteration of a large mesh runs its kernels on overlapping 62-column strips (apply_tile<.., OV>): their own
    // partial-row layout (more strips per block than the 64-column tiling)
    bool vk_overlap = false;
    std::vector<int> poff_ov;
    int poff_edge_ov = 0, nwg_apply_ov = 0;
    std::vector<int> poff;          // partial-row offset of each owned block's K2 launch
    int poff_edge = 0, nwg_apply = 0, nwg_vec = 0;
    std::vector<int> poff2, rows2;  // same for the fused two-sweep launches, and their rows per workgroup
    int poff2_edge = 0, nwg_apply2 = 0;
    std::vector<int> poff3, rows3;  // ... and for the three-sweep launches (all perimeter rows fixed)
    int nwg_apply3 = 0;
    bool fuse_triples = false;
    // Coupled triples (single process): K2x3 with a frozen perimeter stores everything but the nodes within two of a side whose
    // perimeter rows move; three perimeter-row passes evaluate the perimeter and that zone level by level (rows within 4 / 3 / 2 nodes)
    bool triples_coupled = false;
    EdgeRowsDev edge_L[3];
    std::vector<EdgeRun> runs_L[3];          // host copies of the three level tables' runs (strip plan of the fused level kernel)
    bool levels_fused = false;               // the three level passes of a coupled triple in one launch (k_edge_levels3; TM_LEVELS_FUSED=0: three launches)
    FusedLevelsDev fused_levels;
    double* d_rhs_L[3] = {nullptr, nullptr, nullptr};
    std::vector<int32_t> order_L[3];
    std::vector<PlanRow> zone_rows[3];   // KIND_INTERIOR rows of the zones (level 1: distance <= 4, level 2: <= 3, level 3: <= 2)
    bool pipelined_single = false;   // single process, coupled blocks: perimeter-row passes on the chain's queue beside the interior pass
    // halo exchange
    int32_t* d_send_ids = nullptr;
    double2* d_send_buf = nullptr;
    int64_t n_send = 0;
    // white control function
    bool white = false;
    ConnShifts white_le{};
    uint64_t outer_done = 0;
    // measurement: HIP event pairs around K2 launches
    int profile = 0;                // 0 = off, k > 0 = every k-th launch of the dominant kernel is bracketed
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
    uint64_t prof_launches = 0;     // event pairs that count as a launch of the dominant kernel (the parts of a split K2x2 pass count once)
    uint64_t prof_timed = 0;        // ... of which carried an event pair
    uint64_t prof_open = 0;         // launches inside the group bracket that is open (see profiled)
    uint64_t prof_phase = 0;        // position within the bracketed / unbracketed alternation
    void profile_close(hipStream_t on);
    void profile_read(double* ms_total, uint64_t* launches, uint64_t* timed);

    void create(const tm_mesh_desc* mesh, const tm_solver_opt* o, const tm_control_fn* c, const tm_comm_hooks* h, void* strm,
                bool measure);
    void upload(const tm_mesh_desc* mesh);
    void download(const tm_mesh_desc* mesh);
    void iterate(uint64_t iterations, tm_stats* stats);
    bool iterate_until(uint64_t max_iterations, double tol, tm_stats* stats);          // true = the scaled residual reached tol
    bool iterate_until_update(uint64_t max_iterations, double tol, tm_stats* stats);   // true = the update of the last outer iteration was <= tol
    double stop_tol = 0.0;          // > 0: a Picard iteration whose start residual is already <= stop_tol returns without solving
    void apply_host(const double* in_xy, double* out_xy, int scaled);
    // the reference's assembled system for the current coordinates (tm_smoother_assemble_csr) and A in through it, in CSR order
    // (tm_smoother_apply_reference_order); single-process handles
    uint64_t assemble_csr_host(int32_t* Ap, int32_t* Ai, double* Ax_x, double* Ax_y, uint64_t nnz_capacity);
    void apply_reference_host(const double* in_xy, double* out_xy);
    struct AssembledCsr {            // device copy, built on demand, dropped by the next iterate / upload
        int32_t *p = nullptr, *i = nullptr;
        double *vx = nullptr, *vy = nullptr;
        uint64_t nnz = 0;
        std::vector<int32_t> h_p, h_i;
    } csr;
    void csr_build_pattern();
    void csr_fill_values();
    void csr_release();
    void rhs_host(double* rhs_xy);
    void control_function_host(double* pq);
    void export_soa_host(int64_t block, double* x, double* y, double* p, double* q);
    void* export_buf = nullptr;     // scratch planes of export_soa_host (hipMalloc, grown on demand)
    size_t export_bytes = 0;

    // building blocks
    void exchange(double2* vec, hipStream_t on = nullptr);   // start (and, without a split hook, finish) the halo exchange of `vec`; on = the handle's stream unless given
    void exchange_finish(hipStream_t on = nullptr);          // split hooks: make the stream wait for the transfer started by exchange()
    // second stream of a multi-rank relax handle: halo exchanges and perimeter rows of a sweep pair run here, beside the interior pass
    hipStream_t side = nullptr;
    hipEvent_t ev_to_side = nullptr, ev_to_main = nullptr, ev_inside[2] = {nullptr, nullptr};
    void fence(hipStream_t from, hipStream_t to, hipEvent_t ev);
    // How the two queues of a pipelined pass (interior pass on the handle's stream, chain on `side`) are ordered against each other.
    // Counters in device memory + one-wave announce / wait kernels need the two streams on DIFFERENT hardware queues (a waiter whose
    // producer sits behind it in the same in-order queue never sees it start); HIP does not promise that, so the handle finds out: once,
    // when `side` is created, one announce-and-wait round in both directions with a limit of milliseconds (queue_self_test).  Events
    // (barrier packets, ~10 us per hop, no assumption) are used when the test fails, when TM_PAIR_SYNC=events asks for them, and when
    // several multi-rank handles share the process.
    enum { ORDER_UNDECIDED = -1, ORDER_COUNTERS = 0, ORDER_EVENTS_REQUESTED = 1, ORDER_EVENTS_SHARED_PROCESS = 2, ORDER_EVENTS_SELF_TEST = 3 };
    int queue_ordering = ORDER_UNDECIDED;
    void ensure_side();              // creates `side` and its events, runs the self-test (first use)
    bool queue_self_test();          // true = the two streams ran side by side
    bool use_counters();             // the decision for the pass being enqueued
    bool transport_warm = false;     // the first exchange of a handle runs alone and synchronised (connections are set up inside it)
    void warm_transport();
    void relax_pairs_pipelined(uint64_t npairs, bool want_partials_last);
    uint32_t* sync_flags = nullptr;   // device counters of the cross-queue dependencies of a sweep pair (k_queue_signal / k_queue_wait)
    uint32_t* h_flags = nullptr;      // pinned
    bool flags_pending = false;
    bool counted = false;             // this handle is one of g_multirank_handles
    bool exchange_pending = false;
    // step >= 0: the Krylov scalar update that consumes the fused dot products follows the reduction (one launch without hooks)
    void apply(const double2* in, double2* out, int mode, int dot, const double2* aux, const double2* xk, double omega, int step = -1);
    void reduce(int nwg);   // partials -> red (+ all-reduce)
    void reduce_update(int nwg, int step, double rtol = 0.0, double atol = 0.0);   // reduce + Krylov scalar update
    // lazy scalar steps (small single-process meshes): see LazyScalars in tm_kernels.h
    int apply_rows = 0;             // K2 rows per chunk of this handle (0 = per-block rule)
    bool fuse_s = false;            // k_apply_vk<VK_S> / k_xr_update_vs instead of k_s_update + apply + k_xr_update
    bool fuse_p = false;            // k_apply_vk<VK_P> instead of k_p_update + apply; p and v alternate with p_alt, v_alt
    bool fuse2 = false;             // two kernels per iteration: k_apply_vk<VK_R> + k_apply_vk<VK_S2>; r alternates with r_alt
    double2 *p_alt = nullptr, *v_alt = nullptr, *r_alt = nullptr;
    void apply_virtual(int kind, const double2* in, const double2* in2, const double2* in3, double2* pout, double2* out, const double2* in4 = nullptr,
                       double2* rout = nullptr, double2* uio = nullptr);
    bool lazy = false;
    double* part_buf[3] = {nullptr, nullptr, nullptr};
    int part_rot = 0;
    KrylovScalars* S_buf[2] = {nullptr, nullptr};
    LazyStep pending[2];
    int npending = 0;
    void flush_pending();
    LazyScalars scalars_for();
    void white_launch(int update);
    void sync();
    void ensure_tmp();

    // GMRES(30) (TM_INNER_GMRES; csrc/tm_gmres.hip): the m + 1 basis vectors, the device-resident Hessenberg / rotation state, a pinned copy
    double2* gm_V = nullptr;
    GmresScalars* gm_S = nullptr;
    GmresScalars* h_gm = nullptr;

   private:
    int picard_bicgstab(tm_stats& st);
    int picard_gmres(tm_stats& st);
    int picard_solve(tm_stats& st) { return opt.inner == TM_INNER_GMRES ? picard_gmres(st) : picard_bicgstab(st); }
    void relax_sweeps(uint64_t n, tm_stats& st);
    void relax_pair(bool want_partials);
    std::vector<int> relax3_rows_of_owned_blocks() const;
    void relax_triple(bool want_partials);
    void relax_triples_coupled(uint64_t ntriples, bool want_partials_last);
    void profiled(const std::function<void()>& launch, bool counts = true, hipStream_t on = nullptr);
    void relax2_launch(int subset, bool counts, int dot, hipStream_t on = nullptr, const QueueWait* wait = nullptr);
};

// Defaults of the inner solve that depend on the size of the system (tm_solver_opt.rtol == 0, max_inner == 0).
// The stop test bounds the scaled RESIDUAL; the distance of the Picard iterate from the exact-solve iterate -- what north_star's
// 1e-10 RMS is about -- is conditioning x residual, and cond(D^-1 A) of the frozen Winslow system grows like the node count.
// Measured with the diagonal-only BiCGStab against the sparse-LU oracle on perturbed n^2 blocks (tests/test_gpu_parity_ladder.py,
// tools/dev/picard_ladder.py): rms error ~ 4e-17 .. 1e-16 x nodes x (rtol / 1e-14) -- 2e-11 at 1025^2,
// 6.5e-11 at 2049^2 and 6e-10 at 4096^2 with rtol 1e-14.  Hence rtol = 7.5e-9 / nodes, between 1e-16 and 1e-14: every size lands
// at <= 3e-11, for 0-40 % more inner iterations on the large meshes (the recurrence residual keeps falling; no stagnation seen down
// to 1e-16).  The iteration cap grows with the mesh too: BiCGStab with the diagonal alone needs ~3-5 sqrt(nodes) iterations.
// Will a handle with these options ever run sweep TRIPLES across ranks (the only schedule that needs the depth-3 halo)?  A pure function of
// the options (and of TM_TRIPLES_COUPLED in the environment), identical on every rank of a job.
inline bool triples_wanted(const tm_solver_opt& o, const tm_control_fn& c) {
    if (o.inner != TM_INNER_RELAX || c.kind != TM_CF_LAPLACE || (o.flags & TM_OPT_SINGLE_SWEEP)) return false;
    if (const char* e = std::getenv("TM_TRIPLES_COUPLED")) return std::atoi(e) != 0;
    return true;
}
// the library's own transport: non-negative = the depth its tables were built for (0: depth 2, 1: depth 3 where the topology has it); -1 = foreign hooks
int rccl_hooks_allow_triples(const tm_comm_hooks* h);

// TM_INNER_AUTO: the multigrid-preconditioned solve from this many nodes in the largest block on (include/tm_hip.h)
constexpr uint64_t AUTO_MG_MIN_BLOCK_NODES = 100000;
constexpr uint64_t AUTO_MG_MIN_LONE_BLOCK_NODES = 1000;   // ... and from this many when no connection couples the blocks (tm_smoother.cpp)
// ... unless the cells' aspect ratio VARIES inside a block (standard deviation of log(g11/g22) over the block's nodes above this): the cycle
// smooths with point Jacobi and coarsens a block by one rule, so boundary-layer clustering (the reference's O-grids: 1.4-1.8; the
// uniform synthetic blocks: 0.3) leaves it a poor preconditioner -- T106 / LS89 refined to 0.9 / 1.3 M nodes: 14.5 / 21 s against
// 5.5 / 12 s with the plain solve (tools/dev/o4h_auto_probe.py)
constexpr double AUTO_MG_MAX_ASPECT_SPREAD = 1.0;

inline double default_rtol(double nodes) {
    const double r = 7.5e-9 / (nodes > 1.0 ? nodes : 1.0);
    return r > 1e-14 ? 1e-14 : (r < 1e-16 ? 1e-16 : r);
}
inline uint64_t default_max_inner(double nodes) {
    const double k = 12.0 * std::sqrt(nodes > 1.0 ? nodes : 1.0);
    return k > 10000.0 ? static_cast<uint64_t>(k) : 10000u;   // the reference caps at 1000 (BiCGStab.zig:19) -- with its far looser stop test
}

}  // namespace tmh

struct tm_smoother {
    tmh::Smoother impl;
};
