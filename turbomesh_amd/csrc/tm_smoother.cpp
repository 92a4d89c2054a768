// Device-resident smoother: the Picard outer loop of reference smooth.zig:74-166 with the
// per-iteration work (fill, solve, residual, copy-back) replaced by matrix-free gfx950 kernels.
//
//   TM_INNER_BICGSTAB  every outer iteration freezes the coordinates X^k, solves
//                      A(X^k) X^{k+1} = b(X^k) for both components at once with BiCGStab on the
//                      row-equilibrated operator (recurrences of BiCGStab.zig:279-370, scalars
//                      device-resident, no host round trip inside an inner iteration), then K7
//                      computes the reference's residual and copies back (smooth.zig:112-153).
//   TM_INNER_RELAX     every outer iteration is one fused Jacobi sweep of the nonlinear system.
#include "tm_smoother.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>

namespace tmh {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw TmError(TM_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(x) hip_check((x), #x)

std::atomic<int> g_multirank_handles{0};
tm_log_fn g_log_sink = nullptr;
void* g_log_ctx = nullptr;

// ------------------------------------------------------------------ arena
DeviceArena::~DeviceArena() {
    for (void* p : owned_) (void)hipFree(p);
}
void DeviceArena::use_workspace(void* base, uint64_t bytes) {
    base_ = static_cast<char*>(base);
    cap_ = bytes;
    external_ = true;
}
void DeviceArena::measure_only() { measure_ = true; }
void* DeviceArena::alloc(uint64_t bytes) {
    const uint64_t aligned = (bytes + 255) & ~uint64_t{255};
    if (measure_) {
        used_ += aligned;
        return nullptr;
    }
    if (external_) {
        const uint64_t off = (used_ + 255) & ~uint64_t{255};
        if (off + aligned > cap_) throw TmError(TM_E_MEMORY, "caller workspace too small (see tm_smoother_workspace_bytes)");
        used_ = off + aligned;
        return base_ + off;
    }
    void* p = nullptr;
    if (hipMalloc(&p, aligned ? aligned : 256) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed (" + std::to_string(aligned) + " bytes)");
    owned_.push_back(p);
    used_ += aligned;
    return p;
}

// ------------------------------------------------------------------ helpers
static Topology topology_from_desc(const tm_mesh_desc* m) {
    if (!m || !m->blocks || m->nblocks == 0) throw TmError(TM_E_ARG, "mesh description without blocks");
    if ((m->nconns && !m->conns) || (m->nbcs && !m->bcs)) throw TmError(TM_E_ARG, "null connection / condition array");
    Topology t;
    for (uint64_t b = 0; b < m->nblocks; ++b) {
        t.ni.push_back(static_cast<int64_t>(m->blocks[b].ni));
        t.nj.push_back(static_cast<int64_t>(m->blocks[b].nj));
    }
    auto rng = [](const tm_range& r) {
        return TopoRange{static_cast<int64_t>(r.block), r.side, static_cast<int64_t>(r.start), static_cast<int64_t>(r.end)};
    };
    for (uint64_t c = 0; c < m->nconns; ++c) {
        TopoConn tc;
        tc.r[0] = rng(m->conns[c].r[0]);
        tc.r[1] = rng(m->conns[c].r[1]);
        tc.periodic = m->conns[c].has_periodicity != 0;
        tc.per[0] = m->conns[c].periodicity[0];
        tc.per[1] = m->conns[c].periodicity[1];
        t.conns.push_back(tc);
    }
    for (uint64_t c = 0; c < m->nbcs; ++c) t.bcs.push_back(TopoCond{rng(m->bcs[c].range), m->bcs[c].kind});
    try {
        t.finalize();
    } catch (const PlanError& e) {
        throw TmError(e.code, e.what());
    }
    return t;
}

static void check_desc_matches(const Topology& t, const tm_mesh_desc* m) {
    if (!m || static_cast<int64_t>(m->nblocks) != t.nblocks()) throw TmError(TM_E_SIZE, "mesh description does not match the handle");
    for (int64_t b = 0; b < t.nblocks(); ++b)
        if (static_cast<int64_t>(m->blocks[b].ni) != t.ni[b] || static_cast<int64_t>(m->blocks[b].nj) != t.nj[b])
            throw TmError(TM_E_SIZE, "block sizes do not match the handle");
}

// smooth.zig:220-275: both sides of every connection must coincide within 1e-15
static void connection_data_check(const Topology& t, const tm_mesh_desc* m, const std::vector<int32_t>& owner, int rank) {
    const double abs_tol = 1e-15;
    for (size_t c = 0; c < t.conns.size(); ++c) {
        const TopoConn& conn = t.conns[c];
        if (owner[conn.r[0].block] != rank || owner[conn.r[1].block] != rank) continue;   // remote side: not visible here
        const ConnShifts s = conn_shifts(t, conn);
        const double* a = m->blocks[conn.r[0].block].xy;
        const double* b = m->blocks[conn.r[1].block].xy;
        for (int64_t k = 0; k < s.count; ++k) {
            const int64_t p0 = s.position[0] + k * s.direction[0], p1 = s.position[1] + k * s.direction[1];
            double x0 = a[2 * p0], y0 = a[2 * p0 + 1];
            if (conn.periodic) {
                x0 = x0 + conn.per[0];
                y0 = y0 + conn.per[1];
            }
            if (!(std::fabs(x0 - b[2 * p1]) <= abs_tol && std::fabs(y0 - b[2 * p1 + 1]) <= abs_tol))
                throw TmError(TM_E_MISMATCH, "non matching points for connection " + std::to_string(c) + " point " + std::to_string(k));
        }
    }
}

void Smoother::sync() { HIPCHK(hipStreamSynchronize(stream)); }

// Everything the handle owns outside the arena; also runs when create() throws half-way (pinned buffers already allocated).
Smoother::~Smoother() {
    if (counted) g_multirank_handles.fetch_sub(1);
    if (h_S) (void)hipHostFree(h_S);
    for (int k = 0; k < 2; ++k) {
        if (h_poll[k]) (void)hipHostFree(h_poll[k]);
        if (ev_poll[k]) (void)hipEventDestroy(ev_poll[k]);
    }
    if (h_red) (void)hipHostFree(h_red);
    if (h_gm) (void)hipHostFree(h_gm);
    if (h_flags) (void)hipHostFree(h_flags);
    for (hipEvent_t e : ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_stop) (void)hipEventDestroy(e);
    for (hipStream_t q : mg_streams) {
        (void)hipStreamSynchronize(q);
        (void)hipStreamDestroy(q);
    }
    for (hipEvent_t e : mg_join) (void)hipEventDestroy(e);
    if (mg_fork) (void)hipEventDestroy(mg_fork);
    if (side) {
        (void)hipStreamSynchronize(side);
        (void)hipEventDestroy(ev_to_side);
        (void)hipEventDestroy(ev_to_main);
        (void)hipEventDestroy(ev_inside[0]);
        (void)hipEventDestroy(ev_inside[1]);
        (void)hipStreamDestroy(side);
    }
    if (export_buf) (void)hipFree(export_buf);
    csr_release();
}

// ------------------------------------------------------------------ create
void Smoother::create(const tm_mesh_desc* mesh, const tm_solver_opt* o, const tm_control_fn* c, const tm_comm_hooks* h, void* strm,
                      bool measure) {
    if (!o) throw TmError(TM_E_ARG, "null solver option");
    if (o->tag != TM_SOLVER_HIP)
        throw TmError(TM_E_UNSUPPORTED, "ExternalSolverNotEnabled: libtm_hip serves only solver tag `hip` (gmres/bicgstab/umfpack/petsc stay on the Zig side)");
    if (o->inner != TM_INNER_BICGSTAB && o->inner != TM_INNER_RELAX && o->inner != TM_INNER_MG_BICGSTAB && o->inner != TM_INNER_AUTO && o->inner != TM_INNER_GMRES)
        throw TmError(TM_E_ARG, "unknown inner strategy");
    opt = *o;
    if (opt.flags & TM_OPT_PRECOND_ILU0)
        throw TmError(TM_E_UNSUPPORTED, "ILU(0) needs the assembled matrix: it is served by tm_csr_solve (seam 2); the matrix-free path preconditions with the diagonal or the multigrid cycle");
    if (opt.inner == TM_INNER_AUTO) {   // size-aware choice, from the global topology alone (include/tm_hip.h)
        uint64_t largest = 0;
        if (mesh && mesh->blocks)
            for (uint64_t b = 0; b < mesh->nblocks; ++b) largest = std::max<uint64_t>(largest, mesh->blocks[b].ni * mesh->blocks[b].nj);
        // (blocks that no connection couples are the cycle's home ground at ANY size -- a lone 33^2 block 5.0 against 7.3 ms per three Picard
        // iterations, 200^2 11 against 37, 1024^2 13 against 666 -- whereas across interfaces the block-local cycle leaves the coupling to the
        // Krylov iteration and needs 200-1400 iterations where a lone block needs 25: 8 coupled 256^2 blocks 978 against 323 ms, and only
        // from ~512^2 per block on does it win again, 8 x 512^2 1.6 against 2.1 s, 2 x 2048^2 1.8 against 24.8 s: tools/dev/auto_crossover.py)
        const bool uncoupled = mesh && mesh->nconns == 0;
        bool cycle = largest >= (uncoupled ? AUTO_MG_MIN_LONE_BLOCK_NODES : AUTO_MG_MIN_BLOCK_NODES);
        // ... and cells whose aspect ratio does not vary much inside a block.  Read off the caller's coordinates, which only a single-process
        // handle is sure to have for every block (the ranks of a job must decide alike, so with hooks the sizes decide alone; so does
        // the sizing call, whose answer has to cover whatever create decides later)
        const bool ranks = h != nullptr && h->nranks >= 1 && h->exchange != nullptr && h->allreduce_sum != nullptr;
        if (cycle && !ranks && !measure && mesh && mesh->blocks)
            for (uint64_t b = 0; b < mesh->nblocks && cycle; ++b)
                if (mesh->blocks[b].ni * mesh->blocks[b].nj >= 1024 &&
                    BlockMG::aspect_spread_of(mesh->blocks[b].xy, static_cast<int>(mesh->blocks[b].ni), static_cast<int>(mesh->blocks[b].nj)) > AUTO_MG_MAX_ASPECT_SPREAD)
                    cycle = false;
        opt.inner = cycle ? TM_INNER_MG_BICGSTAB : TM_INNER_BICGSTAB;
    }
    if (!(opt.atol > 0)) opt.atol = 0.0;
    if (opt.check_every == 0) opt.check_every = (opt.inner == TM_INNER_MG_BICGSTAB) ? 1 : 8;   // a multigrid-preconditioned iteration costs ~100x a poll
    if (!(opt.omega > 0)) opt.omega = 1.0;
    cf = c ? *c : tm_control_fn{TM_CF_LAPLACE, 0, 0.0, 0.0};
    if (cf.kind != TM_CF_LAPLACE && cf.kind != TM_CF_WHITE) throw TmError(TM_E_ARG, "unknown control function");
    stream = static_cast<hipStream_t>(strm);

    topo = topology_from_desc(mesh);
    dof_global = topo.dof;
    // size-aware for the diagonal-only solves (default_rtol, tm_smoother.hpp); the multigrid-preconditioned solve leaves a residual with a
    // flat spectrum and sits on the fp64 floor of the exact iterate at EVERY size with 1e-14 (4.9e-12 rms from the
    // 1e-16 run at 4096^2, true residual against the oracle-assembled system on the storage floor) -- tightening it buys iterations only
    if (!(opt.rtol > 0))
        opt.rtol = (opt.flags & TM_OPT_RTOL_INITIAL) ? 1e-2 : (opt.inner == TM_INNER_MG_BICGSTAB ? 1e-14 : default_rtol(static_cast<double>(dof_global)));
    if (opt.max_inner == 0) opt.max_inner = default_max_inner(static_cast<double>(dof_global));
    try {
        all_rows = build_rows(topo);
        has_hooks = h != nullptr && h->nranks >= 1 && h->exchange != nullptr && h->allreduce_sum != nullptr;
        if (h) hooks = *h;
        owner.assign(topo.nblocks(), 0);
        int rank = 0, nranks = 1;
        if (has_hooks) {
            if (!h->owner) throw TmError(TM_E_ARG, "hooks need the block -> rank owner table");
            if (h->rank < 0 || h->rank >= h->nranks) throw TmError(TM_E_ARG, "hooks: rank outside [0, nranks)");
            rank = h->rank;
            nranks = h->nranks;
            owner.assign(h->owner, h->owner + topo.nblocks());
        }
        // depth of the halo: what this handle's schedule needs -- unless the hooks are the library's own transport, whose send / receive
        // tables were built when the hooks were made and rule (tm_rccl_hooks: by topology; tm_rccl_hooks_for: by the options)
        bool allow_triples = triples_wanted(opt, cf);
        if (has_hooks) {
            const int t = rccl_hooks_allow_triples(h);
            if (t >= 0) allow_triples = t != 0;
        }
        lp = build_local_plan(topo, all_rows, owner, rank, nranks, allow_triples);
    } catch (const PlanError& e) {
        throw TmError(e.code, e.what());
    }
    if (has_hooks && !measure && !counted) {
        counted = true;
        g_multirank_handles.fetch_add(1);
    }
    n_owned = lp.n_owned;
    n_ghost = static_cast<int64_t>(lp.ghost_gid.size());
    n_local = n_owned + n_ghost;
    if (n_owned == 0) throw TmError(TM_E_ARG, "this rank owns no block");

    white = cf.kind == TM_CF_WHITE;
    if (white) {   // wall_control_function.zig:72, 204-217: hard-coded to blocks 0,1 and connection 0
        if (topo.nblocks() < 2 || topo.conns.empty()) throw TmError(TM_E_TOPOLOGY, "white control function needs the O4H layout (blocks 0,1 + connection 0)");
        const TopoConn& c0 = topo.conns[0];
        if (!(c0.r[0].block == 0 && c0.r[0].start == 0 && c0.r[0].side == SIDE_J_MIN && c0.r[1].block == 1 && c0.r[1].start == 0 &&
              c0.r[1].side == SIDE_J_MIN && !c0.periodic))
            throw TmError(TM_E_TOPOLOGY, "white control function: connection 0 must join blocks 0 and 1 at j_min, start 0, without periodicity");
        if (owner[0] != lp.rank || owner[1] != lp.rank) throw TmError(TM_E_UNSUPPORTED, "white control function: blocks 0 and 1 must live on the same rank");
        white_le = conn_shifts(topo, c0);
    }

    if (measure) arena.measure_only();
    else if (h && h->workspace) arena.use_workspace(h->workspace, h->workspace_bytes);

    // ---- vectors
    auto vec = [&]() { return arena.alloc_n<double2>(static_cast<uint64_t>(n_local)); };
    X = vec();
    U = vec();
    use_mg = opt.inner == TM_INNER_MG_BICGSTAB;
    // the perimeter step of the preconditioner (precondition()): meshes with connections or sliding rows (all-fixed perimeters have identity rows:
    // nothing to apply); decided by the global topology, so every rank of a job decides alike (across ranks the step takes an exchange of the
    // corrections inside every application -- microseconds beside two cycles)
    mg_perimeter_step = use_mg && (!topo.conns.empty() || !topo.bcs.empty());
    if (const char* e = std::getenv("TM_MG_PERIMETER_STEP")) mg_perimeter_step = mg_perimeter_step && std::atoi(e) != 0;
    if (const char* e = std::getenv("TM_MG_PERIMETER_SWEEPS")) mg_perimeter_sweeps = std::max(1, std::atoi(e));
    mg_dirichlet = mg_perimeter_step;   // ... and the perimeter values as Dirichlet data in front of the cycles (block-local: no exchange)
    if (const char* e = std::getenv("TM_MG_DIRICHLET")) mg_dirichlet = mg_dirichlet && std::atoi(e) != 0;
    if (opt.inner == TM_INNER_GMRES) {   // w / z of GMRES.zig:27-38 in one vector, the basis v_0 .. v_m contiguous behind it
        r = vec();
        gm_V = arena.alloc_n<double2>(static_cast<uint64_t>(n_local) * (GMRES_M + 1));
        gm_S = arena.alloc_n<GmresScalars>(1);
    }
    if (opt.inner == TM_INNER_BICGSTAB || use_mg) {
        r = vec();
        r_hat = vec();
        p = vec();
        v = vec();
        s = vec();
        t = vec();
    }
    if (white) PQ = vec();
    if (use_mg) {
        p_hat = vec();
        s_hat = vec();
        mg_w0 = vec();
        mg_w1 = vec();
        mg.resize(lp.owned_blocks.size());
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
            const int64_t b = lp.owned_blocks[k];
            const int bi = static_cast<int>(topo.ni[b]), bj = static_cast<int>(topo.nj[b]);
            const double* xy = (mesh && mesh->blocks) ? mesh->blocks[b].xy : nullptr;
            mg[k].build(arena, bi, bj, white, BlockMG::aspect_of(xy, bi, bj), measure);
        }
        // Small and medium blocks: every level of a cycle is a handful of launch-bound kernels, and the cycles of different blocks share nothing
        // -- run them side by side, a stream per block (the reference's examples refined 8 x: 8 blocks, ~190 us of dependent launches each
        // per cycle).  Blocks that fill the device on their own gain nothing and keep the handle's stream.
        if (!measure && lp.owned_blocks.size() > 1 && lp.n_owned / static_cast<int64_t>(lp.owned_blocks.size()) <= 1500000) {
            int ns = static_cast<int>(std::min<size_t>(lp.owned_blocks.size(), 8));
            if (const char* e = std::getenv("TM_MG_STREAMS")) ns = std::max(0, std::min(ns, std::atoi(e)));
            if (ns > 1) {
                mg_streams.resize(ns);
                mg_join.resize(ns);
                for (int q = 0; q < ns; ++q) {
                    HIPCHK(hipStreamCreateWithFlags(&mg_streams[q], hipStreamNonBlocking));
                    HIPCHK(hipEventCreateWithFlags(&mg_join[q], hipEventDisableTiming));
                }
                HIPCHK(hipEventCreateWithFlags(&mg_fork, hipEventDisableTiming));
            }
        }
    }
    // two sweeps per pass (K2x2): Laplace control function only (White updates P,Q between sweeps), every owned block >= 5 x 5
    fuse_pairs = opt.inner == TM_INNER_RELAX && !white && !(opt.flags & TM_OPT_SINGLE_SWEEP);
    { const char* e = std::getenv("TM_PAIR_SYNC"); pair_sync_events = e && std::strcmp(e, "events") == 0; }
    // (every block of the mesh, not just the owned ones: the ranks of a job must agree on the schedule -- a pair costs one exchange)
    for (int64_t b = 0; b < topo.nblocks(); ++b)
        if (has_hooks || owner[b] == lp.rank) fuse_pairs = fuse_pairs && relax2_supported(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]));
    if (fuse_pairs) M = vec();
    {   // coupled triples: a single process (no neighbouring rank), large blocks, some perimeter row that moves
        bool any_nf = false;
        for (const PlanRow& pr : lp.rows) any_nf = any_nf || pr.kind != KIND_FIXED;
        // ... or several ranks whose blocks are all large enough for the exchange of a depth-3 halo once per triple (LocalPlan::triple_halo,
        // decided from the topology alone: every rank with a neighbour takes the same schedule)
        const bool peers = has_hooks && (!lp.ghost_gid.empty() || !lp.send_ids.empty());
        // a single process with coupled blocks: triples (two queues, the level passes in one launch) at every size -- round 3 had them from 2^20
        // owned nodes on; with the fused level kernel (tools/dev/triples_single_threshold.sh, us per sweep triples / pairs): 8 x 128^2 6.8 / 9.1,
        // 8 x 256^2 6.1 / 10.2, 2 x 724^2 5.8 / 10.2, 8 x 512^2 10.8 / 13.2
        int64_t single_min = 0;
        if (const char* e = std::getenv("TM_TRIPLES_SINGLE_MIN_NODES")) single_min = std::atoll(e);
        triples_coupled = fuse_pairs && any_nf && (peers ? lp.triple_halo : lp.n_owned >= single_min);
        for (int64_t b : lp.owned_blocks) triples_coupled = triples_coupled && topo.ni[b] >= 16 && topo.nj[b] >= 16 && relax3_supported(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]));
        if (const char* e = std::getenv("TM_TRIPLES_COUPLED")) triples_coupled = triples_coupled && std::atoi(e) != 0;
        if (triples_coupled) M2 = vec();
    }
    {   // K2x2's result stores (launch_relax2_block): plain while the rank's field is a good fraction of the 256 MB Infinity Cache but not
        // more than it holds beside the field being read -- measured crossovers: 1024^2 (16 MiB) streaming, 1448^2 .. 2896^2 plain, 4096^2 streaming
        const double mib = 16.0 * static_cast<double>(lp.n_owned) / (1024.0 * 1024.0);
        relax2_store_nt = !(mib > 24.0 && mib <= 160.0);
        if (const char* e = std::getenv("TM_R2_STORE_NT")) relax2_store_nt = std::atoi(e) != 0;
    }
    // Interior pass of a multi-rank sweep pair: when the chain through the border (perimeter rows -> border workgroups -> perimeter
    // rows -> exchange) is longer than the interior pass -- blocks of a few million nodes -- its short kernels must not queue for
    // wave slots: at the START of an interior pass every slot is taken and none is given back for ~10 us.  Asking for a third of
    // the CU's 160 KB of LDS caps the interior pass at 3 workgroups per CU (its registers allow 4): 19.9 -> 18.2 us per sweep at
    // 2048^2; at 4096^2 the interior pass is the longer one and the cap costs 3 % (58.4 -> 60.3), so it is not applied there.
    // ... unless the fields live in the Infinity Cache (plain result stores, above): there the pass is not what starves the chain's
    // kernels, and both the cap and its streaming loads cost more than they give (2048^2, rank 1 of 3: 18.2-19.1 -> 16.1-16.6 us per sweep)
    if (fuse_pairs && has_hooks && lp.n_owned <= 6 * 1024 * 1024 && relax2_store_nt) inside_lds = 54 * 1024;
    if (const char* e = std::getenv("TM_INSIDE_LDS_KB")) inside_lds = static_cast<size_t>(std::max(0, std::atoi(e))) * 1024;

    // ---- perimeter rows -> device SoA with rank-local ids
    const size_t nr = lp.rows.size();
    h_rhs.assign(nr * 2, 0.0);
    auto up = [&](const void* src, uint64_t bytes) -> void* {
        void* d = arena.alloc(bytes);
        if (!measure && bytes) HIPCHK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        return d;
    };
    // Per-row table -> runs (tm_kernels.h EdgeRun).  Rows are grouped by everything that must be equal along a run (static fields
    // and the grid line of their block they lie on), sorted by row id within a group and cut wherever an index stops advancing by
    // the stride of the stretch.  order[p] = position in `sel` of the row whose right-hand side sits at position p of e.rhs.
    // (keep != nullptr: the host copy of the runs and, per run, where it lies -- block, direction, which half of the block, position of
    //  its first point along the line and the position step -- for the strip plan of the fused level kernel)
    struct RunWhere {
        int64_t side_key;            // (block, runs along rows or columns, lower or upper half of the block)
        int32_t pos0, pos_stride;
    };
    auto build_table = [&](const std::vector<const PlanRow*>& sel, EdgeRowsDev& e, double*& rhs_dev, std::vector<int32_t>& order,
                           std::vector<EdgeRun>* keep = nullptr, std::vector<RunWhere>* keep_where = nullptr) {
        const size_t n = sel.size();
        struct HostRow {
            int32_t row, col[9], met[4];
            uint8_t flags;
            int64_t line;   // (block, grid line) the node lies on: rows of different lines never share a run
            int64_t side_key;
            int32_t pos;
        };
        std::vector<HostRow> hr(n);
        auto loc = [&](int64_t gid) {
            const int64_t l = lp.to_local(gid);
            if (l < 0) throw TmError(TM_E_TOPOLOGY, "internal: column is neither owned nor ghost");
            return static_cast<int32_t>(l);
        };
        for (size_t k = 0; k < n; ++k) {
            const PlanRow& pr = *sel[k];
            HostRow& h = hr[k];
            std::memset(&h, 0, sizeof(h));
            h.row = loc(pr.gid);
            h.flags = pr.flags;
            // a ghost copy of a row whose rhs is the node's own boundary coordinate takes it from the row's current value
            if (h.row >= lp.n_owned) h.flags |= static_cast<uint8_t>((pr.rhs_coord & 3) << 2);
            for (int q = 0; q < pr.ncols; ++q) h.col[q] = loc(pr.col[q]);
            if (pr.kind == KIND_SMOOTHED)
                for (int q = 0; q < 4; ++q) h.met[q] = loc(pr.metric[q]);
            int64_t b = topo.nblocks() - 1;
            while (pr.gid < topo.start[b]) --b;
            const int64_t flat = pr.gid - topo.start[b], bi = flat / topo.nj[b], bj = flat % topo.nj[b];
            bool on_row = bi <= 1 || bi >= topo.ni[b] - 2;
            if (pr.kind == KIND_INTERIOR) {   // zone rows of the coupled triples: runs along the nearer pair of sides
                on_row = std::min(bi, topo.ni[b] - 1 - bi) <= std::min(bj, topo.nj[b] - 1 - bj);
                if (h.row < lp.n_owned) h.flags |= 16;
            }
            h.line = (b << 34) | (static_cast<int64_t>(on_row ? 0 : 1) << 33) | (on_row ? bi : bj);
            const int64_t across = on_row ? bi : bj, across_n = on_row ? topo.ni[b] : topo.nj[b];
            h.side_key = (b << 2) | (static_cast<int64_t>(on_row ? 0 : 1) << 1) | (2 * across >= across_n ? 1 : 0);
            h.pos = static_cast<int32_t>(on_row ? bj : bi);
        }
        auto same_static = [&](size_t x, size_t y) {
            const PlanRow &p = *sel[x], &q = *sel[y];
            return p.kind == q.kind && p.ncols == q.ncols && p.self == q.self && hr[x].flags == hr[y].flags && hr[x].line == hr[y].line &&
                   std::memcmp(p.slot, q.slot, sizeof(p.slot)) == 0 && std::memcmp(p.cx, q.cx, sizeof(p.cx)) == 0 &&
                   std::memcmp(p.cy, q.cy, sizeof(p.cy)) == 0 && std::memcmp(p.per, q.per, sizeof(p.per)) == 0;
        };
        auto static_less = [&](size_t x, size_t y) {   // any strict weak order that is consistent with same_static
            const PlanRow &p = *sel[x], &q = *sel[y];
            if (hr[x].line != hr[y].line) return hr[x].line < hr[y].line;
            if (p.kind != q.kind) return p.kind < q.kind;
            if (p.ncols != q.ncols) return p.ncols < q.ncols;
            if (p.self != q.self) return p.self < q.self;
            if (hr[x].flags != hr[y].flags) return hr[x].flags < hr[y].flags;
            int c = std::memcmp(p.slot, q.slot, sizeof(p.slot));
            if (c) return c < 0;
            c = std::memcmp(p.cx, q.cx, sizeof(p.cx));
            if (c) return c < 0;
            c = std::memcmp(p.cy, q.cy, sizeof(p.cy));
            if (c) return c < 0;
            c = std::memcmp(p.per, q.per, sizeof(p.per));
            if (c) return c < 0;
            return hr[x].row < hr[y].row;
        };
        std::vector<size_t> idx(n);
        for (size_t k = 0; k < n; ++k) idx[k] = k;
        std::sort(idx.begin(), idx.end(), static_less);
        std::vector<EdgeRun> runs;
        order.clear();
        for (size_t p = 0; p < n;) {
            const size_t k0 = idx[p];
            const PlanRow& pr = *sel[k0];
            EdgeRun R;
            std::memset(&R, 0, sizeof(R));
            R.first = static_cast<int32_t>(p);
            R.count = 1;
            R.row0 = hr[k0].row;
            R.kind = pr.kind;
            R.ncols = pr.ncols;
            R.self = pr.self;
            R.flags = hr[k0].flags;
            std::memcpy(R.slot, pr.slot, sizeof(R.slot));
            std::memcpy(R.cx, pr.cx, sizeof(R.cx));
            std::memcpy(R.cy, pr.cy, sizeof(R.cy));
            std::memcpy(R.per, pr.per, sizeof(R.per));
            for (int q = 0; q < 9; ++q) R.col0[q] = hr[k0].col[q];
            for (int q = 0; q < 4; ++q) R.met0[q] = hr[k0].met[q];
            size_t e_ = p + 1;
            if (e_ < n && same_static(k0, idx[e_])) {   // strides from the second row, then as far as they hold
                const HostRow& h1 = hr[idx[e_]];
                R.row_stride = h1.row - R.row0;
                for (int q = 0; q < 9; ++q) R.col_stride[q] = h1.col[q] - R.col0[q];
                for (int q = 0; q < 4; ++q) R.met_stride[q] = h1.met[q] - R.met0[q];
                auto fits = [&](size_t j) {
                    if (!same_static(k0, idx[j])) return false;
                    const HostRow& h = hr[idx[j]];
                    const int32_t kk = static_cast<int32_t>(j - p);
                    if (h.row != R.row0 + kk * R.row_stride) return false;
                    for (int q = 0; q < 9; ++q)
                        if (h.col[q] != R.col0[q] + kk * R.col_stride[q]) return false;
                    for (int q = 0; q < 4; ++q)
                        if (h.met[q] != R.met0[q] + kk * R.met_stride[q]) return false;
                    return true;
                };
                while (e_ < n && fits(e_)) ++e_;
                R.count = static_cast<int32_t>(e_ - p);
            }
            for (size_t j = p; j < e_; ++j) order.push_back(static_cast<int32_t>(idx[j]));
            runs.push_back(R);
            if (keep_where) keep_where->push_back(RunWhere{hr[k0].side_key, hr[k0].pos, e_ - p > 1 ? hr[idx[p + 1]].pos - hr[k0].pos : 0});
            p = e_;
        }
        if (keep) *keep = runs;
        std::vector<int32_t> wg_run, wg_k0;
        for (size_t r = 0; r < runs.size(); ++r)
            for (int32_t k0 = 0; k0 < runs[r].count; k0 += EDGE_BLOCK) {
                wg_run.push_back(static_cast<int32_t>(r));
                wg_k0.push_back(k0);
            }
        if (std::getenv("TM_DEBUG_RUNS")) {
            int singles = 0;
            for (const EdgeRun& R : runs) singles += R.count == 1;
            std::fprintf(stderr, "[tm] perimeter-row table: %zu rows -> %zu runs (%d of one row), %zu workgroups\n", n, runs.size(), singles, wg_run.size());
        }
        e.nrows = static_cast<int>(n);
        e.nwg = static_cast<int>(wg_run.size());
        e.runs = static_cast<const EdgeRun*>(up(runs.data(), runs.size() * sizeof(EdgeRun)));
        e.wg_run = static_cast<const int32_t*>(up(wg_run.data(), wg_run.size() * 4));
        e.wg_k0 = static_cast<const int32_t*>(up(wg_k0.data(), wg_k0.size() * 4));
        rhs_dev = arena.alloc_n<double>(n * 2);
        e.rhs = rhs_dev;
    };
    std::vector<const PlanRow*> all(nr);
    for (size_t k = 0; k < nr; ++k) all[k] = &lp.rows[k];
    build_table(all, edge, d_rhs, order_all);
    // Relaxation sweeps never have to touch a `fixed` row: it returns its boundary coordinate (smooth.zig:790-795), which the
    // perimeter of every field buffer holds from upload() on.  They run the perimeter-row kernel over the other rows only
    // (none at all for a block with fixed walls), and the K2x2 workgroups along sides without such rows do not have to wait
    // for it.  dyn_mask: bit 0 = row i = 0 has non-fixed rows, 1 = row ni-1, 2 = column j = 0, 3 = column nj-1.
    nf_rows.clear();
    dyn_mask.assign(lp.owned_blocks.size(), 0);
    for (size_t k = 0; k < nr; ++k) {
        const PlanRow& pr = lp.rows[k];
        if (pr.kind == KIND_FIXED) continue;
        nf_rows.push_back(k);
        int64_t b = topo.nblocks() - 1;
        while (pr.gid < topo.start[b]) --b;
        const size_t kb = std::lower_bound(lp.owned_blocks.begin(), lp.owned_blocks.end(), b) - lp.owned_blocks.begin();
        const int64_t flat = pr.gid - topo.start[b], bi = flat / topo.nj[b], bj = flat % topo.nj[b];
        // a corner node counts for its ROW only (the tiles along that row include the corner tile): the end points of an interface
        // along i = 0 must not turn the two side walls into sides whose workgroups wait for the perimeter-row pass
        const bool corner_row = bi == 0 || bi == topo.ni[b] - 1;
        if (bi == 0) dyn_mask[kb] |= 1;
        if (bi == topo.ni[b] - 1) dyn_mask[kb] |= 2;
        if (bj == 0 && !corner_row) dyn_mask[kb] |= 4;
        if (bj == topo.nj[b] - 1 && !corner_row) dyn_mask[kb] |= 8;
    }
    if (opt.inner == TM_INNER_RELAX) {
        std::vector<const PlanRow*> sel;
        for (size_t k : nf_rows) sel.push_back(&lp.rows[k]);
        build_table(sel, edge_nf, d_rhs_nf, order_nf);
        // multi-rank sweep pairs: the same rows plus the depth-1 ghost rows, evaluated one sweep ahead (LocalPlan::ghost_rows)
        if (fuse_pairs && has_hooks && !lp.ghost_rows.empty()) {
            for (const PlanRow& g : lp.ghost_rows) sel.push_back(&g);
            build_table(sel, edge_nf_g, d_rhs_nf_g, order_nf_g);
        }
        if (triples_coupled) {
            // level l (1..3) evaluates the moving perimeter rows and, as KIND_INTERIOR rows (K2's own arithmetic on the gathered 3 x 3
            // neighbourhood), the interior nodes within 5 - l of a side whose perimeter rows move (Chebyshev distance: the 9-point
            // stencil's dependency cone); what level l reads at level l - 1 lies within 6 - l of such a side or on the perimeter
            std::vector<RunWhere> where_L3;
            for (int lev = 0; lev < 3; ++lev) {
                const int64_t depth = 4 - lev;
                zone_rows[lev].clear();
                for (size_t kb = 0; kb < lp.owned_blocks.size(); ++kb) {
                    const int64_t b = lp.owned_blocks[kb], bi_n = topo.ni[b], bj_n = topo.nj[b];
                    const int dyn = dyn_mask[kb];
                    if (!dyn) continue;
                    auto add = [&](int64_t i, int64_t j) {
                        PlanRow r{};
                        r.gid = topo.start[b] + i * bj_n + j;
                        r.kind = KIND_INTERIOR;
                        r.ncols = 9;
                        r.self = 4;
                        int q = 0;
                        for (int64_t di = -1; di <= 1; ++di)
                            for (int64_t dj = -1; dj <= 1; ++dj) r.col[q++] = r.gid + di * bj_n + dj;
                        zone_rows[lev].push_back(r);
                    };
                    for (int64_t i = 1; i <= bi_n - 2; ++i) {   // ascending gid
                        const bool row_in = ((dyn & 1) && i <= depth) || ((dyn & 2) && i >= bi_n - 1 - depth);
                        if (row_in) {
                            for (int64_t j = 1; j <= bj_n - 2; ++j) add(i, j);
                            continue;
                        }
                        if (dyn & 4)
                            for (int64_t j = 1; j <= depth; ++j) add(i, j);
                        if (dyn & 8)
                            for (int64_t j = bj_n - 1 - depth; j <= bj_n - 2; ++j) add(i, j);
                    }
                }
                std::vector<const PlanRow*> sl;
                for (size_t k : nf_rows) sl.push_back(&lp.rows[k]);
                for (const PlanRow& z : zone_rows[lev]) sl.push_back(&z);
                // several ranks: level 1 also evaluates every row of the depth-2 ghost set, level 2 the depth-1 rows (their owners'
                // definitions, hence their owners' bits); level 3 reads them
                if (lev == 0) for (const PlanRow& g : lp.ghost_rows2) sl.push_back(&g);
                if (lev == 1) for (const PlanRow& g : lp.ghost_rows) sl.push_back(&g);
                build_table(sl, edge_L[lev], d_rhs_L[lev], order_L[lev], &runs_L[lev], lev == 2 ? &where_L3 : nullptr);
            }
            // ---- strip plan of the fused level kernel (k_edge_levels3): level-3 rows by strips of LEVEL_STRIP positions along their lines,
            // per side of a block; per strip the level-2 rows its level-3 rows read, and the level-1 rows THOSE read (hulls per run)
            levels_fused = true;
            if (const char* e = std::getenv("TM_LEVELS_FUSED")) levels_fused = std::atoi(e) != 0;
            if (levels_fused) {
                int LEVEL_STRIP = 60;   // positions per strip: 60 + 2 x 2 halo positions make whole 64-point wave tasks at level 1
                if (const char* e = std::getenv("TM_LEVEL_STRIP")) LEVEL_STRIP = std::max(4, std::atoi(e));
                std::unordered_map<int32_t, std::pair<int32_t, int32_t>> made[2];   // local row id -> (run, k) in the level-1 / level-2 table
                for (int lev = 0; lev < 2; ++lev)
                    for (size_t r = 0; r < runs_L[lev].size(); ++r)
                        for (int32_t k = 0; k < runs_L[lev][r].count; ++k)
                            made[lev][runs_L[lev][r].row0 + k * runs_L[lev][r].row_stride] = {static_cast<int32_t>(r), k};
                auto reads = [&](const EdgeRun& R, int32_t k, auto&& f) {   // every local id row k of run R reads at the previous level
                    for (int q = 0; q < R.ncols; ++q) f(R.col0[q] + k * R.col_stride[q]);
                    if (R.kind == KIND_SMOOTHED)
                        for (int q = 0; q < 4; ++q) f(R.met0[q] + k * R.met_stride[q]);
                    f(R.row0 + k * R.row_stride);
                };
                using Hull = std::map<int32_t, std::pair<int32_t, int32_t>>;   // run -> [kmin, kmax]
                auto widen = [](Hull& h, int32_t run, int32_t k) {
                    auto it = h.find(run);
                    if (it == h.end()) h[run] = {k, k};
                    else {
                        it->second.first = std::min(it->second.first, k);
                        it->second.second = std::max(it->second.second, k);
                    }
                };
                std::map<std::pair<int64_t, int32_t>, Hull> strips;   // (side, strip number) -> level-3 rows
                for (size_t r = 0; r < runs_L[2].size(); ++r)
                    for (int32_t k = 0; k < runs_L[2][r].count; ++k)
                        widen(strips[{where_L3[r].side_key, (where_L3[r].pos0 + k * where_L3[r].pos_stride) / LEVEL_STRIP}], static_cast<int32_t>(r), k);
                std::vector<LevelTask> tasks;
                std::vector<int32_t> off;
                auto emit = [&](const Hull& h) {
                    for (const auto& kv : h)
                        for (int32_t k = kv.second.first; k <= kv.second.second; k += 64)
                            tasks.push_back(LevelTask{kv.first, k, std::min<int32_t>(64, kv.second.second - k + 1)});
                };
                for (const auto& st : strips) {
                    // a strip's level-3 rows need not be one range per run (two strips of one run are separate map entries: they are);
                    // level 2 = hull of what they read and level 2 makes, level 1 = hull of what THAT hull reads and level 1 makes
                    Hull h2, h1;
                    for (const auto& kv : st.second)
                        for (int32_t k = kv.second.first; k <= kv.second.second; ++k)
                            reads(runs_L[2][kv.first], k, [&](int32_t id) {
                                auto it = made[1].find(id);
                                if (it != made[1].end()) widen(h2, it->second.first, it->second.second);
                            });
                    for (const auto& kv : h2)
                        for (int32_t k = kv.second.first; k <= kv.second.second; ++k)
                            reads(runs_L[1][kv.first], k, [&](int32_t id) {
                                auto it = made[0].find(id);
                                if (it != made[0].end()) widen(h1, it->second.first, it->second.second);
                            });
                    off.push_back(static_cast<int32_t>(tasks.size()));
                    emit(h1);
                    off.push_back(static_cast<int32_t>(tasks.size()));
                    emit(h2);
                    off.push_back(static_cast<int32_t>(tasks.size()));
                    emit(st.second);
                    off.push_back(static_cast<int32_t>(tasks.size()));
                }
                fused_levels.nstrips = static_cast<int>(strips.size());
                fused_levels.tasks = static_cast<const LevelTask*>(up(tasks.data(), tasks.size() * sizeof(LevelTask)));
                fused_levels.off = static_cast<const int32_t*>(up(off.data(), off.size() * 4));
                if (std::getenv("TM_DEBUG_RUNS"))
                    std::fprintf(stderr, "[tm] fused level kernel: %d strips, %zu wave tasks (level tables: %d / %d / %d rows)\n", fused_levels.nstrips, tasks.size(),
                                 edge_L[0].nrows, edge_L[1].nrows, edge_L[2].nrows);
            }
        }
    }

    // ---- reductions
    // K2 chunk length for this handle: all owned blocks go into one launch -- the shortest chunks (multiples of the 3-row load
    // group) that keep interior + perimeter workgroups within what the lazy scalar steps allow (512 partial rows); large meshes:
    // the per-block rule of tm_kernels.hip (apply_rows = 0)
    const int edge_wg = (opt.inner == TM_INNER_RELAX ? edge_nf.nwg : edge.nwg);   // a relax handle only ever launches the non-fixed rows
    apply_rows = 0;
    auto launch_wgs = [&](int rows, bool overlap) {   // workgroups of one interior + perimeter launch with chunks of `rows` (0 = the per-block rule)
        int total = edge_wg;
        for (int64_t b : lp.owned_blocks)
            total += overlap ? apply_block_nwg_overlap(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), rows)
                             : apply_block_nwg(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), rows);
        return total;
    };
    // (the two-kernel BiCGStab iteration looks further, up to 36 rows: staying within 512 workgroups keeps its scalar steps lazy, which is worth
    // more than short chunks once the launch fills the device anyway -- single blocks 1500^2 / 1700^2 / 1900^2: 139 -> 121, 163 -> 144,
    // 198 -> 189 us per iteration, T106 refined 8 x 95 -> 90.5; every other path keeps the rule it was tuned with)
    const bool vk_able = !has_hooks && opt.inner == TM_INNER_BICGSTAB && !(opt.flags & TM_OPT_EAGER_SCALARS);
    auto shortest_rows = [&](bool overlap) {
        for (int r = 3; r <= (vk_able ? 36 : 15); r += 3)
            if (launch_wgs(r, overlap) <= 512) return r;
        return 0;
    };
    apply_rows = shortest_rows(false);
    // Layout of the two kernels of the two-kernel BiCGStab iteration (see vk_overlap below, where its tables are built): overlapping
    // 62-column strips between 50 000 and 12 million nodes -- 128^2 .. 724^2 and 1200^2 .. 2896^2 -6 .. -12 % per iteration, the reference's examples
    // T106 / LS89 26.6 -> 22.9 / 28.4 -> 24.0 us, the same meshes refined 2 .. 8 x (8 blocks, O-grid blocks of 81 columns) -22 .. -29 %
    // (tools/dev/vk_overlap_sizes.py, o4h_refined_trace.py) -- with ONE exception, 1024^2 (72 -> 79 us): four 256-column workgroups per
    // chunk row fit its 1024 columns exactly and their 228 workgroups find a CU each, the 62-column strips need five per chunk row and
    // 285.  So: unless the launch with halo loads fits one workgroup per CU and the one with overlapping strips does not.
    // (from 50 000 nodes on: below, the layouts are 3-4 us of ~25 per iteration apart, and the reference's own examples -- 25 / 38 k nodes, whose
    // late Picard iterates amplify every rounding difference a hundredfold, DESIGN.md section 2 -- keep the layout their parity figures were taken with)
    vk_overlap = vk_able && lp.n_owned >= 50000 && lp.n_owned <= 12000000;
    if (vk_overlap && launch_wgs(apply_rows, false) <= 256 && launch_wgs(shortest_rows(true), true) > 256) vk_overlap = false;
    if (const char* e = std::getenv("TM_VK_OVERLAP")) vk_overlap = vk_able && std::atoi(e) != 0;   // forces it off / on (A/B runs, tests)
    if (vk_overlap) apply_rows = shortest_rows(true);   // chunks for the layout that runs (never shorter than the other layout's, which has fewer workgroups per chunk row: both stay within the 512 partial rows of the lazy scalar steps)
    // ... and when no height fits: ONE height for all blocks of a multi-block mesh instead of the per-block rule, which hands blocks of 10^4 nodes
    // 3-row chunks (5 rows read per 3) beside 18-row chunks of the large ones -- LS89 refined 8 x 124.5 -> 110.2, T106 12 x 153.0 -> 130.7,
    // LS89 12 x 206.6 -> 204.9 us per iteration with 24 rows (18: 124.0 / 150.2 / 185.1; 36: 130.2 / 142.4 / 193.0)
    if (vk_able && !apply_rows && lp.owned_blocks.size() > 1) apply_rows = 24;
    // a relax handle launches K2 for single sweeps only (one per pass: TM_OPT_SINGLE_SWEEP, the odd sweep behind pairs / triples): that
    // pass is bandwidth-bound and prefers chunks of ONE six-row load group -- 4096^2: 90.8 against 96.8 us with 18 rows (0.74 against
    // 0.69 of the HBM peak), 2048^2 28.1 / 28.6, 1024^2 8.1 / 8.6 (tools/dev/steady_time.py, STEADY_SINGLE=1); the Krylov kernels,
    // with their seven streams, keep 18 (853 us per iteration against 874-990 for 12 / 6 / 30 / 42 rows)
    if (!apply_rows && opt.inner == TM_INNER_RELAX) apply_rows = 6;
    if (const char* e = std::getenv("TM_APPLY_ROWS")) apply_rows = std::max(0, std::atoi(e));
    poff.clear();
    int off = 0;
    for (int64_t b : lp.owned_blocks) {
        poff.push_back(off);
        off += apply_block_nwg(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), apply_rows);
    }
    poff_edge = off;
    off += edge_wg;
    nwg_apply = off;
    if (fuse_pairs) {
        poff2.clear();
        rows2.clear();
        int off2 = 0;
        for (int64_t b : lp.owned_blocks) {
            const int bi = static_cast<int>(topo.ni[b]), bj = static_cast<int>(topo.nj[b]);
            poff2.push_back(off2);
            rows2.push_back(relax2_rows_per_chunk(bi, bj));
            off2 += relax2_block_nwg(bi, bj, rows2.back());
            const std::vector<int32_t> ids = relax2_border_tiles(bi, bj, rows2.back(), dyn_mask[poff2.size() - 1]);
            border_n.push_back(static_cast<int>(ids.size()));
            border_ids.push_back(static_cast<const int32_t*>(up(ids.data(), ids.size() * 4)));
        }
        poff2_edge = off2;
        nwg_apply2 = off2 + edge_nf.nwg;
    }
    // A single process with coupled blocks (BASELINE configs[3] on one GPU: 8 x 2048^2, seven interfaces): the two perimeter-row passes
    // of a pair are latency-bound (16 us each between two bandwidth-bound launches, 14 % of the pair) -- run them on the chain's queue
    // beside the interior workgroups like a rank with neighbours does (relax_pairs_pipelined; nothing is exchanged)
    pipelined_single = fuse_pairs && !(has_hooks && (lp.send_ids.size() > 0 || !lp.ghost_gid.empty())) && edge_nf.nrows > 0 && lp.n_owned >= (1 << 21);
    if (const char* e = std::getenv("TM_SINGLE_PIPELINED")) pipelined_single = pipelined_single && std::atoi(e) != 0;
    // three sweeps per pass (K2x3): every perimeter row of every owned block `fixed` (nothing for a perimeter-row kernel to do, no
    // neighbour to exchange with) -- a single block with prescribed walls, independent slices
    fuse_triples = fuse_pairs && edge_nf.nrows == 0 && !(has_hooks && (!lp.send_ids.empty() || !lp.ghost_gid.empty()));
    if (const char* e = std::getenv("TM_FUSE_3")) fuse_triples = fuse_triples && std::atoi(e) != 0;
    if (fuse_triples) {
        poff3.clear();
        rows3.clear();
        int off3 = 0;
        rows3 = relax3_rows_of_owned_blocks();
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
            const int64_t b = lp.owned_blocks[k];
            const int bi = static_cast<int>(topo.ni[b]), bj = static_cast<int>(topo.nj[b]);
            if (!relax3_supported(bi, bj)) fuse_triples = false;
            poff3.push_back(off3);
            off3 += relax3_block_nwg(bi, bj, rows3[k]);
        }
        nwg_apply3 = off3;
    }
    if (triples_coupled) {   // the same K2x3 grid, plus the level-3 perimeter-row pass behind it in the partial sums
        poff3.clear();
        rows3.clear();
        int off3 = 0;
        rows3 = relax3_rows_of_owned_blocks();
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
            const int64_t b = lp.owned_blocks[k];
            const int bi = static_cast<int>(topo.ni[b]), bj = static_cast<int>(topo.nj[b]);
            poff3.push_back(off3);
            off3 += relax3_block_nwg(bi, bj, rows3[k]);
        }
        nwg_apply3 = off3;
    }
    nwg_vec = vec_nwg(n_owned);
    const int nwg3_all = fuse_triples ? nwg_apply3 : (triples_coupled ? nwg_apply3 + std::max(edge_L[2].nwg, fused_levels.nstrips) : 0);
    // Overlapping strips for the two kernels of the two-kernel BiCGStab iteration (single process, no preconditioner -- where fuse2 holds,
    // below): no halo registers -> 234 instead of 276 VGPRs for VK_R, two workgroups per CU without spills.  Where it pays was MEASURED, same
    // box, alternating runs (tools/dev/vk_overlap_sizes.py, us per iteration halo loads / overlapping strips): 128^2 20.1 / 17.9, 512^2 34.3 / 31.8,
    // 724^2 49.8 / 44.3, 1024^2 72-74 / 79-80, 1200^2 97.1 / 86.0, 1448^2 124.7 / 110.7, 2048^2 271.2 / 255.7, 2896^2 516.0 / 485.9,
    // 4096^2 862-882 / 870-918, 5792^2 1610 / 1733: at 4096^2 and beyond the pass is bound by the HBM streams alone and the 3 % of re-read
    // columns cost what the second workgroup per CU gives.  The decision is taken with the chunk heights above.
    if (vk_overlap) {
        int off_ov = 0;
        for (int64_t b : lp.owned_blocks) {
            poff_ov.push_back(off_ov);
            off_ov += apply_block_nwg_overlap(static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), apply_rows);
        }
        poff_edge_ov = off_ov;
        nwg_apply_ov = off_ov + edge.nwg;
    }
    // Krylov modes on a small single-process mesh: the scalar steps travel with the kernels that consume their result (LazyScalars,
    // tm_kernels.h) -- three partial-sum buffers in rotation, two scalar blocks
    // (the two-kernel BiCGStab iteration launches no vector kernel inside its loop: there the interior + perimeter launches alone decide
    // whether the scalar steps can travel lazily, and the few vector kernels around the loop take 512 workgroups -- T106 / LS89 refined 4 x,
    // 0.28 / 0.41 M nodes: 46.2 -> 42.8 / 62.6 -> 59.4 us per iteration, same iterates bit for bit; more than 512 partial rows do NOT pay:
    // 2048 of them in every workgroup's prologue made T106 x 8 115 instead of 95 us)
    if (vk_able && std::max(nwg_apply, nwg_apply_ov) <= 512) nwg_vec = std::min(nwg_vec, 512);
    lazy = !has_hooks && opt.inner != TM_INNER_RELAX && std::max(std::max(nwg_apply, nwg_apply_ov), nwg_vec) <= 512 && !(opt.flags & TM_OPT_EAGER_SCALARS);
    const uint64_t npart = static_cast<uint64_t>(std::max(std::max(std::max(std::max(nwg_apply, nwg_apply_ov), nwg_apply2), nwg3_all), nwg_vec)) * MAX_PARTIALS;
    // second apply of an iteration with the s-update folded in: single process (nothing of s has to travel) and no preconditioner
    // (which wants s as a stored vector)
    fuse_s = !has_hooks && opt.inner == TM_INNER_BICGSTAB && !(opt.flags & TM_OPT_EAGER_SCALARS);
    // ... and the first apply with the p-update folded in: p and v alternate between two arrays each (see k_apply_vk)
    fuse_p = fuse_s;
    if (const char* e = std::getenv("TM_FUSE_P")) fuse_p = fuse_p && std::atoi(e) != 0;
    // ... and the whole iteration in two kernels (see picard_bicgstab): r alternates too
    fuse2 = fuse_p;
    if (const char* e = std::getenv("TM_FUSE_2")) fuse2 = fuse2 && std::atoi(e) != 0;
    if (fuse_p) {
        p_alt = arena.alloc_n<double2>(static_cast<uint64_t>(n_local));
        v_alt = arena.alloc_n<double2>(static_cast<uint64_t>(n_local));
    }
    if (fuse2) r_alt = arena.alloc_n<double2>(static_cast<uint64_t>(n_local));
    for (int k = 0; k < (lazy ? 3 : 1); ++k) part_buf[k] = arena.alloc_n<double>(npart);
    partials = part_buf[0];
    red = arena.alloc_n<double>(MAX_PARTIALS);
    S_buf[0] = arena.alloc_n<KrylovScalars>(1);
    S_buf[1] = lazy ? arena.alloc_n<KrylovScalars>(1) : S_buf[0];
    S = S_buf[0];
    sync_flags = arena.alloc_n<uint32_t>(64);   // [0] border passes done, [1] interior passes done, [2] a wait timed out

    // ---- halo exchange
    n_send = static_cast<int64_t>(lp.send_ids.size());
    if (n_send) {
        d_send_ids = static_cast<int32_t*>(up(lp.send_ids.data(), static_cast<uint64_t>(n_send) * 4));
        d_send_buf = arena.alloc_n<double2>(static_cast<uint64_t>(n_send));
    }
    if (measure) return;

    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_S), sizeof(KrylovScalars), hipHostMallocDefault));
    pipelined_poll = lazy;   // small single-process meshes (see poll_done)
    if (const char* e = std::getenv("TM_PIPELINED_POLL")) pipelined_poll = pipelined_poll && std::atoi(e) != 0;
    if (pipelined_poll)
        for (int k = 0; k < 2; ++k) {
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_poll[k]), sizeof(KrylovScalars), hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&ev_poll[k], hipEventDisableTiming));
        }
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_red), sizeof(double) * MAX_PARTIALS, hipHostMallocDefault));
    if (gm_S) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_gm), sizeof(GmresScalars), hipHostMallocDefault));
        HIPCHK(hipMemsetAsync(gm_S, 0, sizeof(GmresScalars), stream));
        HIPCHK(hipMemsetAsync(gm_V, 0, sizeof(double2) * static_cast<size_t>(n_local) * (GMRES_M + 1), stream));
        HIPCHK(hipMemsetAsync(r, 0, sizeof(double2) * n_local, stream));
    }
    HIPCHK(hipMemsetAsync(S_buf[0], 0, sizeof(KrylovScalars), stream));
    if (S_buf[1] != S_buf[0]) HIPCHK(hipMemsetAsync(S_buf[1], 0, sizeof(KrylovScalars), stream));
    HIPCHK(hipMemsetAsync(sync_flags, 0, sizeof(uint32_t) * 64, stream));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_flags), sizeof(uint32_t) * 8, hipHostMallocDefault));   // [0..3] pairs, [4..7] triples
    std::memset(h_flags, 0, sizeof(uint32_t) * 8);
    for (double2* q : {p_hat, s_hat, mg_w0, mg_w1})
        if (q) HIPCHK(hipMemsetAsync(q, 0, sizeof(double2) * n_local, stream));
    if (PQ) HIPCHK(hipMemsetAsync(PQ, 0, sizeof(double2) * n_local, stream));
    upload(mesh);
    if (white) white_launch(0);   // ControlFunction.init, wall_control_function.zig:27-42
    sync();
}

// ------------------------------------------------------------------ upload / download
void Smoother::upload(const tm_mesh_desc* mesh) {
    check_desc_matches(topo, mesh);
    for (int64_t b : lp.owned_blocks)
        if (!mesh->blocks[b].xy) throw TmError(TM_E_ARG, "owned block without coordinates");
    connection_data_check(topo, mesh, owner, lp.rank);
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        HIPCHK(hipMemcpyAsync(X + lp.local_start[k], mesh->blocks[b].xy, sizeof(double2) * topo.ni[b] * topo.nj[b], hipMemcpyHostToDevice, stream));
    }
    // static right-hand sides: fixed rows keep the boundary coordinates, sliding rows the boundary x (smooth.zig:794-795, 853-858)
    for (size_t k = 0; k < lp.rows.size(); ++k) {
        const PlanRow& pr = lp.rows[k];
        h_rhs[2 * k] = pr.rhs[0];
        h_rhs[2 * k + 1] = pr.rhs[1];
        if (pr.rhs_coord) {
            int64_t b = topo.nblocks() - 1;
            while (pr.gid < topo.start[b]) --b;
            const double* xy = mesh->blocks[b].xy + 2 * (pr.gid - topo.start[b]);
            if (pr.rhs_coord & 1) h_rhs[2 * k] = xy[0];
            if (pr.rhs_coord & 2) h_rhs[2 * k + 1] = xy[1];
        }
    }
    // device right-hand sides are stored in run order (build_table: order[p] = row of the selection at position p)
    std::vector<double> staged;
    auto put = [&](double* dev, const std::vector<int32_t>& order, const std::function<const double*(int32_t)>& rhs_of) {
        if (!dev || order.empty()) return;
        const size_t base = staged.size();
        staged.resize(base + 2 * order.size());
        for (size_t p = 0; p < order.size(); ++p) {
            const double* v = rhs_of(order[p]);
            staged[base + 2 * p] = v[0];
            staged[base + 2 * p + 1] = v[1];
        }
        HIPCHK(hipMemcpyAsync(dev, staged.data() + base, sizeof(double) * 2 * order.size(), hipMemcpyHostToDevice, stream));
    };
    staged.reserve(2 * (order_all.size() + order_nf.size() + order_nf_g.size() + order_L[0].size() + order_L[1].size() + order_L[2].size()));   // no reallocation while copies are in flight
    put(d_rhs, order_all, [&](int32_t k) { return &h_rhs[2 * static_cast<size_t>(k)]; });
    put(d_rhs_nf, order_nf, [&](int32_t k) { return &h_rhs[2 * nf_rows[k]]; });
    // own rows as above, then the ghost rows' static right-hand sides (coordinate parts come from the row's value)
    put(d_rhs_nf_g, order_nf_g, [&](int32_t k) {
        return static_cast<size_t>(k) < nf_rows.size() ? &h_rhs[2 * nf_rows[k]] : lp.ghost_rows[static_cast<size_t>(k) - nf_rows.size()].rhs;
    });
    static const double zero_rhs[2] = {0.0, 0.0};   // interior rows have b = 0
    for (int lev = 0; lev < 3; ++lev)
        put(d_rhs_L[lev], order_L[lev], [&](int32_t k) -> const double* {
            const size_t q = static_cast<size_t>(k), nz = zone_rows[lev].size();
            if (q < nf_rows.size()) return &h_rhs[2 * nf_rows[q]];
            if (q < nf_rows.size() + nz) return zero_rhs;
            const std::vector<PlanRow>& g = lev == 0 ? lp.ghost_rows2 : lp.ghost_rows;   // static parts; coordinate parts come from the row's value
            return g[q - nf_rows.size() - nz].rhs;
        });
    if (opt.inner == TM_INNER_RELAX) prefill_fixed();
    sync();   // host staging buffers may go away after return
}

// Every field buffer of a relax handle carries the boundary coordinates on its perimeter from here on: `fixed` rows are
// never evaluated again (their value IS this coordinate), the other perimeter rows are overwritten sweep by sweep.
void Smoother::prefill_fixed() {
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        const int64_t ls = lp.local_start[k];
        const int bi = static_cast<int>(topo.ni[b]), bj = static_cast<int>(topo.nj[b]);
        HIPCHK(launch_copy_perimeter(X + ls, U + ls, bi, bj, stream));
        if (M) HIPCHK(launch_copy_perimeter(X + ls, M + ls, bi, bj, stream));
        if (M2) HIPCHK(launch_copy_perimeter(X + ls, M2 + ls, bi, bj, stream));
    }
}

void Smoother::download(const tm_mesh_desc* mesh) {
    check_desc_matches(topo, mesh);
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        if (!mesh->blocks[b].xy) throw TmError(TM_E_ARG, "owned block without coordinates");
        HIPCHK(hipMemcpyAsync(mesh->blocks[b].xy, X + lp.local_start[k], sizeof(double2) * topo.ni[b] * topo.nj[b], hipMemcpyDeviceToHost, stream));
    }
    sync();
}

// ------------------------------------------------------------------ building blocks
void Smoother::exchange(double2* vec, hipStream_t on) {
    if (!has_hooks || (n_send == 0 && n_ghost == 0)) return;
    hipStream_t st = on ? on : stream;
    const double2* send = vec;   // direct: every peer's rows are one contiguous run of the vector (exchange plan offsets = local ids)
    if (!lp.direct_send) {
        HIPCHK(launch_gather_rows(vec, d_send_ids, n_send, d_send_buf, st));
        send = d_send_buf;
    }
    const int rc = hooks.exchange(hooks.ctx, reinterpret_cast<const double*>(send), reinterpret_cast<double*>(vec + n_owned), st);
    if (rc != 0) throw TmError(TM_E_COMM, "halo exchange hook failed with code " + std::to_string(rc));
    exchange_pending = hooks.exchange_wait != nullptr;
}

void Smoother::exchange_finish(hipStream_t on) {
    if (!exchange_pending) return;
    exchange_pending = false;
    const int rc = hooks.exchange_wait(hooks.ctx, on ? on : stream);
    if (rc != 0) throw TmError(TM_E_COMM, "halo exchange wait hook failed with code " + std::to_string(rc));
}

void Smoother::fence(hipStream_t from, hipStream_t to, hipEvent_t ev) {
    HIPCHK(hipEventRecord(ev, from));
    HIPCHK(hipStreamWaitEvent(to, ev, 0));
}

void Smoother::reduce(int nwg) {
    HIPCHK(launch_finalize(partials, nwg, red, stream));
    if (has_hooks) {
        const int rc = hooks.allreduce_sum(hooks.ctx, red, MAX_PARTIALS, stream);
        if (rc != 0) throw TmError(TM_E_COMM, "all-reduce hook failed with code " + std::to_string(rc));
    }
}

void Smoother::reduce_update(int nwg, int step, double rtol, double atol) {
    if (has_hooks) {   // the sums of all ranks are needed before the scalars can move
        reduce(nwg);
        HIPCHK(launch_scalar_update(S, red, step, stream, rtol, atol));
        return;
    }
    if (lazy && step != STEP_TOL) {   // no launch: the step waits for the kernel that needs its result (scalars_for)
        if (npending == 2) flush_pending();
        pending[npending++] = LazyStep{step, partials, nwg};
        part_rot = (part_rot + 1) % 3;
        partials = part_buf[part_rot];   // the next producer writes elsewhere: this buffer is read by the consumer's workgroups
        return;
    }
    flush_pending();
    HIPCHK(launch_finalize_scalar(partials, nwg, red, S, step, stream, rtol, atol));
}

// pending steps applied by their own launches (before the host reads the scalars, or a step that cannot wait)
void Smoother::flush_pending() {
    for (int q = 0; q < npending; ++q) HIPCHK(launch_finalize_scalar(pending[q].partials, pending[q].nwg, red, S, pending[q].step, stream));
    npending = 0;
}

// the scalars for a kernel that reads them: with pending steps, the kernel applies them itself and publishes the result to the
// other scalar block, which is the current one from then on
LazyScalars Smoother::scalars_for() {
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
}

// runs `launch`; with profiling on, bracketed by a hipEvent pair on the handle's stream
void Smoother::profiled(const std::function<void()>& launch, bool counts, hipStream_t on) {
    if (!on) on = stream;
    if (!profile) {
        launch();
        return;
    }
    auto next_events = [&]() {
        if (ev_used == ev_start.size()) {
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0));
            HIPCHK(hipEventCreate(&e1));
            ev_start.push_back(e0);
            ev_stop.push_back(e1);
        }
    };
    // A handle whose passes follow each other with nothing in between (single process, no perimeter rows to evaluate: a block with
    // fixed walls) is timed in GROUPS: one event pair around `profile` consecutive launches, every other group -- an event record is
    // a barrier packet whose latency (several us here) would otherwise be charged to every single launch it brackets.
    if (!has_hooks && edge_nf.nrows == 0 && counts && profile > 1) {
        const uint64_t G = static_cast<uint64_t>(profile), pos = prof_phase % (2 * G);
        if (pos == 0) {
            next_events();
            HIPCHK(hipEventRecord(ev_start[ev_used], on));
            prof_open = 0;
        }
        launch();
        prof_launches += 1;
        prof_phase += 1;
        if (pos < G) {
            prof_open += 1;
            if (pos == G - 1) profile_close(on);
        }
        return;
    }
    // sampling: only every profile-th pass carries event pairs (two extra packets on the stream per pair).  The parts of a
    // split pass (inside: counts == false, then border: counts == true) see the same prof_launches, hence the same decision.
    const bool prof_sample = (prof_launches % static_cast<uint64_t>(profile)) == 0;
    if (!prof_sample) {
        launch();
        if (counts) prof_launches += 1;
        return;
    }
    next_events();
    HIPCHK(hipEventRecord(ev_start[ev_used], on));
    launch();
    HIPCHK(hipEventRecord(ev_stop[ev_used], on));
    ev_used += 1;
    if (counts) {
        prof_launches += 1;
        prof_timed += 1;
    }
}

// ends a group bracket (see profiled): after its last launch, or when the call that opened it runs out of launches
void Smoother::profile_close(hipStream_t on) {
    if (prof_open == 0) return;
    HIPCHK(hipEventRecord(ev_stop[ev_used], on ? on : stream));
    ev_used += 1;
    prof_timed += prof_open;
    prof_open = 0;
}

void Smoother::apply(const double2* in, double2* out, int mode, int dot, const double2* aux, const double2* xk, double omega, int step) {
    exchange(const_cast<double2*>(in));
    std::vector<ApplyBlock> blocks(lp.owned_blocks.size());
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        const int64_t ls = lp.local_start[k];
        ApplyBlock& a = blocks[k];
        a.in = in + ls;
        a.xk = xk + ls;
        a.pq = PQ ? PQ + ls : nullptr;
        a.aux = aux ? aux + ls : nullptr;
        a.out = out + ls;
        a.ni = static_cast<int>(topo.ni[b]);
        a.nj = static_cast<int>(topo.nj[b]);
        a.omega = omega;
        a.rows = apply_rows;
        a.partials = partials + static_cast<size_t>(poff[k]) * MAX_PARTIALS;
    }
    // Small meshes on a single-process handle: interior rows of all blocks AND the perimeter rows in one launch (no exchange has to
    // land in between) -- they are bound by dependent launches, and the two passes are each a few memory latencies long.
    if (!has_hooks && nwg_apply <= 1024 && !profile) {
        const hipError_t rc = launch_apply_edge_blocks(blocks.data(), static_cast<int>(blocks.size()), mode, dot, edge, in, xk, PQ, aux, out,
                                                       partials + static_cast<size_t>(poff_edge) * MAX_PARTIALS, stream);
        if (rc == hipSuccess) {
            if (dot != DOT_NONE && dot != DOT_DELTA) {
                if (step >= 0) reduce_update(nwg_apply, step);
                else reduce(nwg_apply);
            }
            return;
        }
        if (rc != hipErrorNotSupported) HIPCHK(rc);
        (void)hipGetLastError();
    }
    // all owned blocks in one launch (groups of APPLY_BATCH_MAX): small multi-block meshes are launch-bound
    profiled([&]() { HIPCHK(launch_apply_blocks(blocks.data(), static_cast<int>(blocks.size()), mode, dot, stream)); });
    exchange_finish();   // K2 above read owned rows only; the perimeter rows below read the ghost rows
    // a relaxation sweep evaluates only the perimeter rows that are not `fixed` (see prefill_fixed)
    HIPCHK(launch_edge_rows(mode == MODE_RELAX ? edge_nf : edge, in, xk, PQ, aux, out, omega, mode, dot, partials + static_cast<size_t>(poff_edge) * MAX_PARTIALS, stream));
    if (dot != DOT_NONE && dot != DOT_DELTA) {   // relax sweeps leave the per-workgroup partials; summed when read
        if (step >= 0) reduce_update(nwg_apply, step);
        else reduce(nwg_apply);
    }
}

// An apply with the vector update(s) in front of it folded in (single-process handles without a preconditioner): see k_apply_vk.
//   VK_S:  out = D^-1 A (in - alpha in2)                                          -> STEP_SS_TSTT
//   VK_P:  out = D^-1 A p', p' = in + beta (in2 - omega in3) stored to pout; r_hat . out -> STEP_SIGMA
//   VK_R:  the pending x / r update, p' and out = D^-1 A p' (in = r, in2 = v, in3 = t, in4 = p)  -> STEP_A2
//   VK_S2: VK_S with the sums that also give the next rho                          -> STEP_B2
void Smoother::apply_virtual(int kind, const double2* in, const double2* in2, const double2* in3, double2* pout, double2* out, const double2* in4, double2* rout,
                             double2* uio) {
    const bool dot_aux = kind == VK_P || kind == VK_R || kind == VK_S2;
    const bool ov = vk_overlap && (kind == VK_R || kind == VK_S2);
    std::vector<ApplyBlock> blocks(lp.owned_blocks.size());
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        const int64_t ls = lp.local_start[k];
        ApplyBlock& a = blocks[k];
        a.in = in + ls;
        a.in2 = in2 + ls;
        a.in3 = in3 ? in3 + ls : nullptr;
        a.in4 = in4 ? in4 + ls : nullptr;
        a.pout = pout ? pout + ls : nullptr;
        a.rout = rout ? rout + ls : nullptr;
        a.uio = uio ? uio + ls : nullptr;
        a.xk = X + ls;
        a.pq = PQ ? PQ + ls : nullptr;
        a.aux = dot_aux ? r_hat + ls : nullptr;
        a.out = out + ls;
        a.ni = static_cast<int>(topo.ni[b]);
        a.nj = static_cast<int>(topo.nj[b]);
        a.omega = 0.0;
        a.rows = apply_rows;
        a.partials = partials + static_cast<size_t>(ov ? poff_ov[k] : poff[k]) * MAX_PARTIALS;
    }
    VirtualIn V;
    V.kind = kind;
    V.in = in;
    V.in2 = in2;
    V.in3 = in3;
    V.in4 = in4;
    V.aux = dot_aux ? r_hat : nullptr;
    V.pout = pout;
    V.rout = rout;
    V.uio = uio;
    HIPCHK(launch_apply_virtual(blocks.data(), static_cast<int>(blocks.size()), edge, V, X, PQ, out,
                                partials + static_cast<size_t>(ov ? poff_edge_ov : poff_edge) * MAX_PARTIALS, scalars_for(), stream, ov));
    reduce_update(ov ? nwg_apply_ov : nwg_apply, kind == VK_P ? STEP_SIGMA : (kind == VK_S ? STEP_SS_TSTT : (kind == VK_R ? STEP_A2 : STEP_B2)));
}

void Smoother::white_launch(int update) {
    WhiteArgs w;
    const auto idx = [&](int64_t b) { return lp.local_start[std::lower_bound(lp.owned_blocks.begin(), lp.owned_blocks.end(), b) - lp.owned_blocks.begin()]; };
    w.x0 = X + idx(0);
    w.x1 = X + idx(1);
    w.pq0 = PQ + idx(0);
    w.pq1 = PQ + idx(1);
    w.ni0 = static_cast<int>(topo.ni[0]);
    w.nj0 = static_cast<int>(topo.nj[0]);
    w.ni1 = static_cast<int>(topo.ni[1]);
    w.nj1 = static_cast<int>(topo.nj[1]);
    w.le_p0 = static_cast<int>(white_le.position[0]);
    w.le_p1 = static_cast<int>(white_le.position[1]);
    w.le_fi0 = static_cast<int>(white_le.first_internal[0]);
    w.le_fi1 = static_cast<int>(white_le.first_internal[1]);
    w.le_dir0 = static_cast<int>(white_le.direction[0]);
    w.ds_target = cf.ds_target;
    w.theta_target = cf.theta_target;
    HIPCHK(launch_white(w, update, stream));
}

// out = M^-1 in: one V-cycle per owned block on the interior rows, identity on the perimeter rows (tm_multigrid.hpp)
void Smoother::precondition(const double2* in, double2* out) {
    double2* const in_w = const_cast<double2*>(in);   // (its first interior ring is changed for the duration of the cycles and restored to the bit)
    const size_t nb = lp.owned_blocks.size(), ns = mg_streams.size();
    auto dims = [&](size_t k, int& bi, int& bj) {
        const int64_t b = lp.owned_blocks[k];
        bi = static_cast<int>(topo.ni[b]);
        bj = static_cast<int>(topo.nj[b]);
    };
    int bi, bj;
    // the perimeter values as Dirichlet data of the blocks' cycles: f_I - (D^-1 A)_Ip f_p on the first interior ring (originals parked in t)
    if (mg_dirichlet)
        for (size_t k = 0; k < nb; ++k) {
            dims(k, bi, bj);
            const int64_t ls = lp.local_start[k];
            HIPCHK(launch_ring_dirichlet(in_w + ls, X + ls, PQ ? PQ + ls : nullptr, t + ls, bi, bj, stream));
        }
    if (ns > 1) HIPCHK(hipEventRecord(mg_fork, stream));   // everything the cycles read is in front of this in the handle's queue
    for (size_t k = 0; k < nb; ++k) {
        const int64_t ls = lp.local_start[k];
        hipStream_t q = ns > 1 ? mg_streams[k % ns] : stream;
        if (ns > 1 && k < ns) HIPCHK(hipStreamWaitEvent(q, mg_fork, 0));
        mg[k].vcycle(in + ls, out + ls, mg_w0 + ls, mg_w1 + ls, q);   // (leaves the perimeter of out zero)
    }
    for (size_t q = 0; q < ns && ns > 1; ++q) {
        HIPCHK(hipEventRecord(mg_join[q], mg_streams[q]));
        HIPCHK(hipStreamWaitEvent(stream, mg_join[q], 0));
    }
    for (size_t k = 0; k < nb; ++k) {
        dims(k, bi, bj);
        const int64_t ls = lp.local_start[k];
        if (mg_dirichlet) HIPCHK(launch_ring_restore(in_w + ls, t + ls, bi, bj, stream));
        if (!mg_perimeter_step) HIPCHK(launch_copy_perimeter(in + ls, out + ls, bi, bj, stream));
    }
    if (!mg_perimeter_step) return;
    // Coupled blocks: the perimeter unknowns are not left with their diagonal alone -- e_p = f_p - (D^-1 A)_pI e_I, the perimeter rows applied to
    // the interior corrections just computed (block upper-triangular instead of block-diagonal: the rows of an interface see the
    // corrections of the first interior rows either side of it).  One perimeter-row launch and a subtraction per block; `t` is free
    // whenever a preconditioner application runs (picard_bicgstab) and lends its perimeter entries.
    // ... and then `mg_perimeter_sweeps - 1` Jacobi sweeps on the perimeter system itself (the rows are equilibrated: unit diagonal), whose
    // couplings ALONG an interface and between the two copies of its nodes the first pass leaves out: with the perimeter rows applied to
    // (e_I, e_p) the same two launches read e_p <- f_p - rows + e_p.  CPU prototype with exact interior solves: 41 / 39 / 54 iterations with
    // the first pass alone, 26 / 28 / 39 with one sweep more, 23 / 28 / 35 with the perimeter system solved exactly.
    for (int sweep = 0; sweep < mg_perimeter_sweeps; ++sweep) {
        if (has_hooks) {   // the rows of an interface read the neighbour's corrections (first interior rows, then perimeter values too): an exchange per pass
            exchange(out);
            exchange_finish();
        }
        HIPCHK(launch_edge_rows(edge, out, X, PQ, nullptr, t, 0.0, MODE_SCALED, DOT_NONE, nullptr, stream));
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
            const int64_t b = lp.owned_blocks[k];
            const int64_t ls = lp.local_start[k];
            HIPCHK(launch_perimeter_sub(in + ls, t + ls, out + ls, static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), stream));
        }
    }
}

// Convergence poll of the inner solve: 0 = keep iterating, 1 = both components converged, 2 = a component broke down.
// A small mesh runs 8 iterations in ~250 us and a blocking read of the scalar block costs ~30 us of idle device: there the copy
// is only ENQUEUED (after iteration `it`), the next iterations follow it into the queue, and the copy made at the PREVIOUS poll
// is the one examined.  Iterations past the end are harmless by construction: a finished component has alpha = omega = beta = 0
// (scalar_update_body), so its U and r no longer move and `done` never reverts.  `poll_iters` = the iteration count of the poll
// that saw the end (what a blocking poll would have reported).
int Smoother::poll_done(uint64_t it, bool final) {
    // Stagnation watch: the default tolerance of the diagonal-only solve lies below the TRUE residual fp64 can reach (it is met
    // by the recurrence residual drifting down, which does keep removing low-frequency error: DESIGN.md section 5); should the recurrence
    // residual of an active component ever fail to halve over max(4000, 4 sqrt(dof)) iterations, the solve is over -- reported as not
    // converged (a warning), without a restart, instead of running to the iteration cap.
    auto verdict = [&](const KrylovScalars* h, uint64_t at) {
        if (h->done[0] && h->done[1]) return (h->done[0] == 1 && h->done[1] == 1) ? 1 : 2;
        bool progress = false;
        for (int c = 0; c < 2; ++c) {
            if (h->done[c]) continue;
            if (!(stall_best[c] > 0.0) || h->rr[c] < 0.25 * stall_best[c]) {   // ||r||^2 down by 4 = the norm halved
                stall_best[c] = h->rr[c];
                progress = true;
            }
        }
        if (progress) stall_since = at;
        else if (at > stall_since && at - stall_since > stall_window) {
            stalled = true;
            return 2;
        }
        return 0;
    };
    flush_pending();
    if (!pipelined_poll) {
        HIPCHK(hipMemcpyAsync(h_S, S, sizeof(KrylovScalars), hipMemcpyDeviceToHost, stream));
        sync();
        poll_iters = it;
        return verdict(h_S, it);
    }
    int v = 0;
    const int cur = poll_slot, prev = poll_slot ^ 1;
    HIPCHK(hipMemcpyAsync(h_poll[cur], S, sizeof(KrylovScalars), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipEventRecord(ev_poll[cur], stream));
    if (poll_open) {
        HIPCHK(hipEventSynchronize(ev_poll[prev]));
        v = verdict(h_poll[prev], poll_it[prev]);
        if (v) poll_iters = poll_it[prev];
    }
    if (!v && final) {
        HIPCHK(hipEventSynchronize(ev_poll[cur]));
        v = verdict(h_poll[cur], it);
        poll_iters = it;
    }
    poll_it[cur] = it;
    poll_open = true;
    poll_slot = prev;
    return v;
}

// ------------------------------------------------------------------ Picard + BiCGStab
// One outer iteration: returns 1 if the inner solve did not converge.
int Smoother::picard_bicgstab(tm_stats& st) {
    if (white && outer_done > 0) white_launch(1);   // system.fill(n): control_function.update for n > 0 (smooth.zig:1107-1110)
    exchange(X);
    exchange_finish();
    if (use_mg)   // level hierarchy of the newly frozen field
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) mg[k].set_field(X + lp.local_start[k], PQ ? PQ + lp.local_start[k] : nullptr, stream);
    // warm start: the solution vector starts from the current coordinates (BiCGStab.zig:136-153; later
    // outer iterations continue from the copied-back solution, which is the same field) -- copied behind the stop test below
    // tolerance from ||D^-1 b||
    HIPCHK(launch_edge_rhs(edge, X, PQ, nullptr, 1, partials, stream));
    reduce_update(edge.nwg, STEP_TOL, (opt.flags & TM_OPT_RTOL_INITIAL) ? -opt.rtol : opt.rtol, opt.atol);

    int restarts = 0;
    uint64_t it_total = 0;
    bool converged = false;
    poll_iters = 0;
    stall_best[0] = stall_best[1] = 0.0;
    stall_since = 0;
    stalled = false;
    stall_window = std::max<uint64_t>(4000, static_cast<uint64_t>(4.0 * std::sqrt(static_cast<double>(dof_global))));
    while (true) {
        // r = D^-1 (b - A U) ; r_hat = r ; p = v = 0   (first pass: U is X)
        apply(restarts == 0 ? X : U, r, MODE_RESID, DOT_OUT2, nullptr, X, 0.0, fuse2 ? STEP_INIT2 : STEP_INIT);
        st.operator_sweeps += 1;
        if (restarts == 0) {   // scaled nonlinear residual of this outer iteration
            flush_pending();
            HIPCHK(hipMemcpyAsync(h_S, S, sizeof(KrylovScalars), hipMemcpyDeviceToHost, stream));
            sync();
            st.scaled_residual_rms = std::sqrt((h_S->rr0[0] + h_S->rr0[1]) / (2.0 * static_cast<double>(dof_global)));
            if (stop_tol > 0.0 && st.scaled_residual_rms <= stop_tol) return 2;   // X already satisfies A(X) X = b to stop_tol: leave it untouched
            if (h_S->done[0] == 1 && h_S->done[1] == 1) {
                HIPCHK(hipMemcpyAsync(U, X, sizeof(double2) * n_local, hipMemcpyDeviceToDevice, stream));   // the copy-back below reads it
                converged = true;
                break;
            }
        }
        // (behind the stop test: an outer iteration that finds its system already solved moves no vector at all)
        if (restarts == 0) HIPCHK(hipMemcpyAsync(U, X, sizeof(double2) * n_local, hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(r_hat, r, sizeof(double2) * n_local, hipMemcpyDeviceToDevice, stream));
        if (fuse2 || fuse_p) {   // the fused forms multiply p and v by beta = 0 in their first pass; k_p_update selects instead
            HIPCHK(hipMemsetAsync(p, 0, sizeof(double2) * n_local, stream));
            HIPCHK(hipMemsetAsync(v, 0, sizeof(double2) * n_local, stream));
        }
        if (fuse2) HIPCHK(hipMemsetAsync(t, 0, sizeof(double2) * n_local, stream));   // the first pass multiplies it by omega = 0
        bool breakdown = false;
        poll_open = false;
        while (it_total < opt.max_inner) {
            if (fuse2) {
                // Two kernels per iteration.  (A) in front of the first apply, as the rows enter its window: the x / r update left
                // pending by the previous iteration (u += alpha p + omega s, r' = s - omega t, s = r - alpha v), then p' = r' +
                // beta (p - omega v) and v' = D^-1 A p'; sums r_hat.v' and ||r'||^2.  (B) the second apply on s' = r' - alpha' v'
                // formed the same way, t = D^-1 A s'; sums t.s, t.t, r_hat.s, r_hat.t -- which give omega AND the next rho
                // (r_hat.r'' = r_hat.s - omega r_hat.t), so no third reduction and no third kernel.  r, p, v alternate between two
                // arrays each: neighbouring workgroups still read the old values in their halos.
                apply_virtual(VK_R, r, v, t, p_alt, v_alt, p, r_alt, U);
                std::swap(r, r_alt);
                std::swap(p, p_alt);
                std::swap(v, v_alt);
                apply_virtual(VK_S2, r, v, nullptr, nullptr, t);
                st.operator_sweeps += 2;
                it_total += 1;
                if (it_total % opt.check_every == 0 || it_total == opt.max_inner) {
                    const int verdict = poll_done(it_total, it_total == opt.max_inner);
                    if (verdict) {
                        converged = verdict == 1;
                        breakdown = !converged;
                        break;
                    }
                }
                continue;
            }
            if (fuse_p) {
                // p = r + beta (p - omega v) is formed as the rows enter the first apply's window and stored from there; the new p
                // and v go to the alternate arrays (the halos of neighbouring workgroups still read the old ones)
                apply_virtual(VK_P, r, p, v, p_alt, v_alt);
                std::swap(p, p_alt);
                std::swap(v, v_alt);
            } else {
                HIPCHK(launch_p_update(scalars_for(), r, p, v, n_owned, stream));
                if (use_mg) {   // right preconditioning (BiCGStab.zig:314-316, 340-342 with M = one V-cycle)
                    precondition(p, p_hat);
                    apply(p_hat, v, MODE_SCALED, DOT_AUX, r_hat, X, 0.0, STEP_SIGMA);
                } else {
                    apply(p, v, MODE_SCALED, DOT_AUX, r_hat, X, 0.0, STEP_SIGMA);
                }
            }
            if (fuse_s) {
                // s = r - alpha v is never stored: the second apply forms it as the rows enter its window (interior rows and
                // perimeter rows), with ||s||^2 beside t.s and t.t in one reduction; k_xr_update_vs forms it again
                apply_virtual(VK_S, r, v, nullptr, nullptr, t);
                HIPCHK(launch_xr_update_vs(scalars_for(), U, p, v, t, r, r_hat, n_owned, partials, stream));
                reduce_update(nwg_vec, STEP_RHO);
                st.operator_sweeps += 2;
                it_total += 1;
                if (it_total % opt.check_every == 0 || it_total == opt.max_inner) {
                    const int verdict = poll_done(it_total, it_total == opt.max_inner);
                    if (verdict) {
                        converged = verdict == 1;
                        breakdown = !converged;
                        break;
                    }
                }
                continue;
            }
            HIPCHK(launch_s_update(scalars_for(), r, v, s, n_owned, partials, stream));
            reduce_update(nwg_vec, STEP_SS);
            if (use_mg) {
                precondition(s, s_hat);
                apply(s_hat, t, MODE_SCALED, DOT_AUX2, s, X, 0.0, STEP_TSTT);   // t.s and t.t with the UNpreconditioned s
            } else {
                apply(s, t, MODE_SCALED, DOT_IN, nullptr, X, 0.0, STEP_TSTT);
            }
            HIPCHK(launch_xr_update(scalars_for(), U, use_mg ? p_hat : p, use_mg ? s_hat : s, s, t, r, r_hat, n_owned, partials, stream));
            reduce_update(nwg_vec, STEP_RHO);
            st.operator_sweeps += 2;
            it_total += 1;
            if (it_total % opt.check_every == 0 || it_total == opt.max_inner) {
                const int verdict = poll_done(it_total, it_total == opt.max_inner);
                if (verdict) {
                    converged = verdict == 1;
                    breakdown = !converged;
                    break;
                }
            }
        }
        if (fuse2) {   // the last iteration's x / r update is still pending (zero scalars once a component has finished)
            flush_pending();
            HIPCHK(launch_xr_update_vs(scalars_for(), U, p, v, t, r, r_hat, n_owned, partials, stream));   // its partial sums are not used
        }
        if (converged || !breakdown || stalled || restarts >= 8 || it_total >= opt.max_inner) break;
        restarts += 1;   // breakdown (rho or omega vanished): restart from the current iterate
    }
    st.inner_iterations += (converged && poll_iters) ? poll_iters : it_total;
    flush_pending();

    // residual + copy-back (smooth.zig:112-153); X becomes the new frozen field
    HIPCHK(launch_residual_copyback(X, U, n_owned, partials, stream));
    reduce(nwg_vec);
    HIPCHK(hipMemcpyAsync(h_red, red, sizeof(double) * MAX_PARTIALS, hipMemcpyDeviceToHost, stream));
    sync();
    st.last_dx2 = h_red[0];
    st.last_dy2 = h_red[1];
    st.last_residual = (h_red[0] + h_red[1]) * (h_red[0] + h_red[1]);   // smooth.zig:136
    outer_done += 1;
    return converged ? 0 : 1;
}

// Picard outer iteration with GMRES(30) as the inner solver (GMRES.zig:300-423 restated on the device, csrc/tm_gmres.hip): left
// preconditioning with the diagonal (GMRES.zig:425-431) = K2's MODE_SCALED, modified Gram-Schmidt with one pass per basis vector, Givens
// rotations and the triangular solve by one-thread kernels on device-resident state, both components together.  Returns like
// picard_bicgstab: 0 converged, 1 not converged (a warning, GMRES.zig:422), 2 stop_tol already met.
int Smoother::picard_gmres(tm_stats& st) {
    if (white && outer_done > 0) white_launch(1);   // system.fill(n): control_function.update for n > 0 (smooth.zig:1107-1110)
    exchange(X);
    exchange_finish();
    HIPCHK(launch_edge_rhs(edge, X, PQ, nullptr, 1, partials, stream));   // ||D^-1 b||^2 -> the tolerance
    reduce(edge.nwg);
    HIPCHK(launch_gm_tol(gm_S, red, (opt.flags & TM_OPT_RTOL_INITIAL) ? -opt.rtol : opt.rtol, opt.atol, stream));
    double2* const W = r;
    const int64_t ld = n_local;
    auto V = [&](int k) { return gm_V + static_cast<int64_t>(k) * ld; };
    auto poll = [&]() {   // the flags of both components, after everything enqueued so far
        HIPCHK(hipMemcpyAsync(h_gm, gm_S, sizeof(GmresScalars), hipMemcpyDeviceToHost, stream));
        sync();
        return h_gm->done[0] == 1 && h_gm->done[1] == 1;
    };
    uint64_t it_total = 0;
    bool converged = false, first = true;
    while (it_total < opt.max_inner) {
        // z = D^-1 (b - A x), beta = ||z|| (GMRES.zig:312-319); the warm start is the current field (GMRES.zig:136-153 seeds x_new from the mesh)
        apply(first ? X : U, W, MODE_RESID, DOT_OUT2, nullptr, X, 0.0, -1);
        st.operator_sweeps += 1;
        HIPCHK(launch_gm_begin(gm_S, red, stream));
        const bool all_done = poll();
        if (first) {
            st.scaled_residual_rms = std::sqrt((h_gm->rr0[0] + h_gm->rr0[1]) / (2.0 * static_cast<double>(dof_global)));
            if (stop_tol > 0.0 && st.scaled_residual_rms <= stop_tol) return 2;
            HIPCHK(hipMemcpyAsync(U, X, sizeof(double2) * n_local, hipMemcpyDeviceToDevice, stream));
            first = false;
        }
        if (all_done) {
            converged = true;
            break;
        }
        HIPCHK(launch_gm_divide(V(0), W, gm_S, n_owned, stream));   // v0 = z / beta
        bool cycle_done = false;
        for (int j = 0; j < GMRES_M && it_total < opt.max_inner; ++j) {
            apply(V(j), W, MODE_SCALED, DOT_NONE, nullptr, X, 0.0, -1);   // w = A v_j, z = D^-1 w (GMRES.zig:335-336)
            st.operator_sweeps += 1;
            // modified Gram-Schmidt (GMRES.zig:338-345): h_ij = z . v_i, z -= h_ij v_i, i = 0 .. j -- the subtraction of step i - 1 and
            // the inner product of step i in one pass each; then h_{j+1,j} = ||z|| behind the last subtraction
            HIPCHK(launch_gm_mgs(W, nullptr, V(0), nullptr, gm_S, 0, n_owned, partials, stream));
            reduce(nwg_vec);
            for (int i = 1; i <= j; ++i) {
                HIPCHK(launch_gm_mgs(W, V(i - 1), V(i), red, gm_S, i - 1, n_owned, partials, stream));
                reduce(nwg_vec);
            }
            HIPCHK(launch_gm_mgs(W, V(j), nullptr, red, gm_S, j, n_owned, partials, stream));
            reduce(nwg_vec);
            HIPCHK(launch_gm_column(gm_S, red, stream));                      // rotations, g, |g_{j+1}|, flags (GMRES.zig:347-386)
            if (j + 1 < GMRES_M + 1) HIPCHK(launch_gm_divide(V(j + 1), W, gm_S, n_owned, stream));   // v_{j+1} = z / h_next unless h_next <= 1e-30
            it_total += 1;
            if ((j + 1) % static_cast<int>(opt.check_every) == 0 || j + 1 == GMRES_M || it_total == opt.max_inner) {
                if (poll()) {
                    cycle_done = true;
                    break;
                }
            }
        }
        HIPCHK(launch_gm_backsub(gm_S, stream));                              // GMRES.zig:394-409
        HIPCHK(launch_gm_update(U, gm_V, ld, gm_S, n_owned, stream));          // x += V y, GMRES.zig:411-417
        if (cycle_done) {
            converged = true;
            break;
        }
    }
    st.inner_iterations += it_total;

    // residual + copy-back (smooth.zig:112-153); X becomes the new frozen field
    HIPCHK(launch_residual_copyback(X, U, n_owned, partials, stream));
    reduce(nwg_vec);
    HIPCHK(hipMemcpyAsync(h_red, red, sizeof(double) * MAX_PARTIALS, hipMemcpyDeviceToHost, stream));
    sync();
    st.last_dx2 = h_red[0];
    st.last_dy2 = h_red[1];
    st.last_residual = (h_red[0] + h_red[1]) * (h_red[0] + h_red[1]);   // smooth.zig:136
    outer_done += 1;
    return converged ? 0 : 1;
}

// Two sweeps in one pass over the interior rows: X^(k+2) = S(S(X^k)), bit-identical to two single sweeps.
//   perimeter rows of X^(k+1)  <- perimeter-row kernel on X^k           (into M)
//   interior rows of X^(k+2)   <- K2x2 (reads X^k and M's perimeter; leaves the first-interior ring of X^(k+1) in M)
//   perimeter rows of X^(k+2)  <- perimeter-row kernel on M (perimeter + ring + exchanged ghost rows are all it reads)
// With several ranks the K2x2 grid is launched in two parts on two queues (relax_pairs_pipelined): the border workgroups in the
// latency-critical chain, the rest beside it.
void Smoother::relax2_launch(int subset, bool counts, int dot, hipStream_t on, const QueueWait* wait) {
    if (!on) on = stream;
    std::vector<Relax2Block> blocks(lp.owned_blocks.size());
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        const int64_t ls = lp.local_start[k];
        Relax2Block& a = blocks[k];
        a.in = X + ls;
        a.mid = M + ls;
        a.out = U + ls;
        a.ni = static_cast<int>(topo.ni[b]);
        a.nj = static_cast<int>(topo.nj[b]);
        a.omega = opt.omega;
        a.dyn = dyn_mask[k];
        a.store_nt = relax2_store_nt ? 1 : 0;
        a.border = border_ids[k];
        a.nborder = border_n[k];
        a.partials = partials + static_cast<size_t>(poff2[k]) * MAX_PARTIALS;
    }
    const size_t lds = (subset == R2_INSIDE) ? inside_lds : 0;
    profiled([&]() { HIPCHK(launch_relax2_blocks(blocks.data(), rows2.data(), static_cast<int>(blocks.size()), dot, subset, on, lds, wait)); }, counts, on);
}

// Rows per chunk of K2x3, chosen per launch: launch_relax3_blocks takes the owned blocks in groups of APPLY_BATCH_MAX
std::vector<int> Smoother::relax3_rows_of_owned_blocks() const {
    std::vector<int> rows(lp.owned_blocks.size(), 1), bi, bj;
    for (int64_t b : lp.owned_blocks) {
        bi.push_back(static_cast<int>(topo.ni[b]));
        bj.push_back(static_cast<int>(topo.nj[b]));
    }
    for (size_t first = 0; first < rows.size(); first += APPLY_BATCH_MAX) {
        const int n = static_cast<int>(std::min<size_t>(APPLY_BATCH_MAX, rows.size() - first));
        relax3_rows_for_launch(bi.data() + first, bj.data() + first, n, rows.data() + first, triples_coupled);
    }
    return rows;
}

// ---- the second queue of a pipelined pass and how it is ordered against the handle's stream (see tm_smoother.hpp)
void Smoother::ensure_side() {
    if (side) return;
    int least = 0, greatest = 0;   // the chain is latency-critical: let its kernels overtake queued interior workgroups
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest));
    HIPCHK(hipEventCreateWithFlags(&ev_to_side, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_to_main, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_inside[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_inside[1], hipEventDisableTiming));
    if (pair_sync_events) queue_ordering = ORDER_EVENTS_REQUESTED;
    else queue_ordering = queue_self_test() ? ORDER_COUNTERS : ORDER_EVENTS_SELF_TEST;
}

// One announce-and-wait round between the handle's stream and `side`, in both directions at once -- the very pair of kernels the triple
// schedule enqueues per triple -- with a limit of 5 ms.  Both streams are drained first, so the only thing that can keep either kernel
// from starting is the other one spinning in front of it in a SHARED hardware queue (a caller's stream of the same priority class as
// `side` with the runtime's per-priority queue pool exhausted, GPU_MAX_HW_QUEUES=1, ...): then the first kernel runs into its limit,
// raises the error word and the handle orders its queues with events from the start.  ~0.1 ms once per handle when it passes.
bool Smoother::queue_self_test() {
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipStreamSynchronize(side));
    uint32_t* a = sync_flags + 8;    // words 8..11: the test's own counters (the passes use 0..3)
    uint32_t* b = sync_flags + 9;
    uint32_t* err = sync_flags + 10;
    HIPCHK(hipMemsetAsync(sync_flags + 8, 0, sizeof(uint32_t) * 4, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(launch_queue_signal_wait(a, b, 1u, err, stream, QUEUE_SELF_TEST_TICKS));
    HIPCHK(launch_queue_signal_wait(b, a, 1u, err, side, QUEUE_SELF_TEST_TICKS));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipStreamSynchronize(side));
    uint32_t h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(h, sync_flags + 8, sizeof(h), hipMemcpyDeviceToHost));
    return h[2] == 0u && h[0] == 1u && h[1] == 1u;
}

bool Smoother::use_counters() {
    if (queue_ordering == ORDER_EVENTS_REQUESTED || queue_ordering == ORDER_EVENTS_SELF_TEST) return false;
    if (g_multirank_handles.load() > 1) {   // several multi-rank handles in ONE process (the virtual-rank tests) wait for each other through shared queues
        queue_ordering = ORDER_EVENTS_SHARED_PROCESS;
        return false;
    }
    queue_ordering = ORDER_COUNTERS;
    return true;
}

// The first exchange of a handle runs alone, with the host waiting for it: a transport sets up its connections inside its first
// transfer (RCCL: seconds, on the host, inside ncclGroupEnd) and the ranks of a job reach their first pass at different times.  Everything
// that skew would otherwise be spent by a device-side wait of the first pass; with it out of the way the waits of a pass only ever cover a
// neighbour's kernels and its host's jitter.  The rows that travel are X's own (the pass sends them again).
void Smoother::warm_transport() {
    if (transport_warm) return;
    transport_warm = true;
    if (!(has_hooks && (n_send > 0 || n_ghost > 0))) return;
    exchange(X, side);
    exchange_finish(side);
    HIPCHK(hipStreamSynchronize(side));
}

// Three sweeps in one pass (K2x3, all perimeter rows fixed): X^(k+3) = S(S(S(X^k))), bit-identical to three single sweeps
void Smoother::relax_triple(bool want_partials) {
    const int dot = want_partials ? DOT_DELTA : DOT_NONE;
    std::vector<Relax2Block> blocks(lp.owned_blocks.size());
    for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
        const int64_t b = lp.owned_blocks[k];
        const int64_t ls = lp.local_start[k];
        Relax2Block& a = blocks[k];
        a.in = X + ls;
        a.mid = nullptr;
        a.out = U + ls;
        a.ni = static_cast<int>(topo.ni[b]);
        a.nj = static_cast<int>(topo.nj[b]);
        a.omega = opt.omega;
        a.dyn = 0;
        a.store_nt = relax2_store_nt ? 1 : 0;
        a.border = nullptr;
        a.nborder = 0;
        a.partials = partials + static_cast<size_t>(poff3[k]) * MAX_PARTIALS;
    }
    profiled([&]() { HIPCHK(launch_relax3_blocks(blocks.data(), rows3.data(), static_cast<int>(blocks.size()), dot, stream)); }, true, stream);
    std::swap(X, U);
}

// Three sweeps per pass on COUPLED blocks of a single process.  K2x3 runs on the whole of every block with the perimeter frozen at
// X^k; three sweeps carry the error of that only two nodes deep along a side whose perimeter rows move, and the pass does not store
// those nodes (Relax2Block::dyn).  Beside it, on the chain's queue, the perimeter-row kernel evaluates level by level what the
// pass cannot: level 1 = the moving perimeter rows of X^(k+1) and the interior nodes within 4 of such a side (from X^k, into M),
// level 2 = perimeter of X^(k+2) and the nodes within 3 (from M into M2), level 3 = perimeter of X^(k+3) and the nodes within 2
// (from M2 into the output).  Interior nodes are evaluated as KIND_INTERIOR rows -- K2's own arithmetic, the same bits -- so the
// result equals three single sweeps bit for bit; the zone is 0.2 % of a 2048^2 block.  One join of the two queues per triple.
// Several ranks (LocalPlan::triple_halo): X^k of the depth-3 ghost set travels once per triple in front of level 1, which also
// evaluates the depth-2 ghost rows, level 2 the depth-1 ones -- with their owners' row definitions, so with their owners' bits.
void Smoother::relax_triples_coupled(uint64_t ntriples, bool want_partials_last) {
    ensure_side();
    // Ordering between the two queues inside the loop: counters in device memory and one-wave signal / wait kernels, as in
    // relax_pairs_pipelined (an event record / wait pair stalls its queue for ~10 us on this part, twice per triple: at 2048^2 that is
    // a third of the interior pass); events when several multi-rank handles share the process or TM_PAIR_SYNC=events asks for them.
    // A waiter's producer is always enqueued before it in host order.
    const bool use_flags = use_counters();
    uint32_t* chain_done = sync_flags;        // level-3 passes completed (side queue)
    uint32_t* inside_done = sync_flags + 1;   // K2x3 passes completed (handle's stream)
    uint32_t* sync_err = sync_flags + 2;
    if (use_flags) HIPCHK(hipMemsetAsync(sync_flags, 0, sizeof(uint32_t) * 4, stream));
    fence(stream, side, ev_to_side);   // X (and the cleared counters) complete on the handle's stream
    warm_transport();
    std::vector<Relax2Block> blocks(lp.owned_blocks.size());
    for (uint64_t q = 0; q < ntriples; ++q) {
        const int dot = (q + 1 == ntriples && want_partials_last) ? DOT_DELTA : DOT_NONE;
        for (size_t k = 0; k < lp.owned_blocks.size(); ++k) {
            const int64_t b = lp.owned_blocks[k];
            const int64_t ls = lp.local_start[k];
            Relax2Block& a = blocks[k];
            a.in = X + ls;
            a.mid = nullptr;
            a.out = U + ls;
            a.ni = static_cast<int>(topo.ni[b]);
            a.nj = static_cast<int>(topo.nj[b]);
            a.omega = opt.omega;
            a.dyn = dyn_mask[k];
            a.store_nt = relax2_store_nt ? 1 : 0;
            a.border = nullptr;
            a.nborder = 0;
            a.partials = partials + static_cast<size_t>(poff3[k]) * MAX_PARTIALS;
        }
        // handle's stream: K2x3 of triple q needs the chain of triple q-1 (its level 3 wrote rows of X).  With counters, ONE one-wave
        // kernel per queue and triple does both jobs: starting behind K2x3 of triple q-1 it announces that pass (inside_done = q), then
        // waits for the chain of triple q-1 (chain_done >= q); its twin on the chain's queue announces the chain and waits for the pass.
        // Both announce before they wait, so neither can hold the other up -- PROVIDED the two streams sit on different hardware queues:
        // on one in-order queue the second could not start while the first spins.  That is not assumed: ensure_side() ran exactly this
        // pair of kernels once when `side` was created (queue_self_test, limit 5 ms) and use_counters() is false when it failed.
        if (q > 0) {
            if (use_flags) HIPCHK(launch_queue_signal_wait(inside_done, chain_done, static_cast<uint32_t>(q), sync_err, stream));
            else HIPCHK(hipStreamWaitEvent(stream, ev_to_main, 0));
        }
        profiled([&]() { HIPCHK(launch_relax3_blocks(blocks.data(), rows3.data(), static_cast<int>(blocks.size()), dot, stream)); }, true, stream);
        if (!use_flags) HIPCHK(hipEventRecord(ev_inside[q & 1], stream));
        // chain: the levels of triple q read X: K2x3 of triple q-1 must be complete (its own level 3 precedes them in this queue)
        if (q > 0) {
            if (use_flags) HIPCHK(launch_queue_signal_wait(chain_done, inside_done, static_cast<uint32_t>(q), sync_err, side));
            else HIPCHK(hipStreamWaitEvent(side, ev_inside[(q - 1) & 1], 0));
        }
        exchange(X, side);                 // several ranks: X^k of the depth-3 ghost set, once per triple (a no-op otherwise)
        exchange_finish(side);
        if (levels_fused) {   // the three level passes in one launch: strips with their own closure of level-1 and level-2 rows (k_edge_levels3)
            HIPCHK(launch_edge_levels3(fused_levels, edge_L[0], edge_L[1], edge_L[2], X, M, M2, U, PQ, opt.omega, dot,
                                       partials + static_cast<size_t>(nwg_apply3) * MAX_PARTIALS, side));
        } else {
            HIPCHK(launch_edge_rows(edge_L[0], X, X, PQ, nullptr, M, opt.omega, MODE_RELAX, DOT_NONE, partials, side));
            HIPCHK(launch_edge_rows(edge_L[1], M, M, PQ, nullptr, M2, opt.omega, MODE_RELAX, DOT_NONE, partials, side));
            HIPCHK(launch_edge_rows(edge_L[2], M2, M2, PQ, nullptr, U, opt.omega, MODE_RELAX, dot, partials + static_cast<size_t>(nwg_apply3) * MAX_PARTIALS, side));
        }
        if (q + 1 < ntriples && !use_flags) HIPCHK(hipEventRecord(ev_to_main, side));
        std::swap(X, U);
    }
    fence(side, stream, ev_to_main);   // the handle's stream continues behind the whole chain
    if (use_flags) {
        HIPCHK(hipMemcpyAsync(h_flags + 4, sync_flags, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, stream));   // (the pairs behind a triple reuse the counters)
        flags_pending = true;
    }
}

// want_partials: only the pass whose displacement norms are read back pays for them (the last one of an iterate() call)
void Smoother::relax_pair(bool want_partials) {
    const int dot = want_partials ? DOT_DELTA : DOT_NONE;
    // (a handle that gets here has no neighbouring rank -- relax_sweeps sends the others through relax_pairs_pipelined: nothing travels)
    HIPCHK(launch_edge_rows(edge_nf, X, X, PQ, nullptr, M, opt.omega, MODE_RELAX, DOT_NONE, partials, stream));
    relax2_launch(R2_ALL, true, dot);
    HIPCHK(launch_edge_rows(edge_nf, M, M, PQ, nullptr, U, opt.omega, MODE_RELAX, dot, partials + static_cast<size_t>(poff2_edge) * MAX_PARTIALS, stream));
    std::swap(X, U);
}

// Sweep pairs of a rank that has neighbours.  What bounds a pair is not bandwidth but the chain of DEPENDENT steps through the
// block's border: border workgroups of pair k -> perimeter rows of X^(k+2) -> halo exchange -> perimeter (+ ghost) rows of
// X^(k+3) -> border workgroups of pair k+1.  A dependency between two queues costs ~13 us on this part (event record + stream
// wait, rocprofv3 trace in profiles/), one inside a queue 1-7 us, so the whole chain lives on ONE high-priority stream (`side`)
// and only the interior pass -- workgroups that touch neither perimeter nor first-interior ring, 88 % of a 4096^2 block, reading
// nothing but interior rows -- runs on the handle's stream beside it.  The two cross-queue waits per pair (interior pass k+1 after
// border k, border k+1 after interior pass k) have a whole pass of slack.
//     side:  x E1g(0) B(0) | E2(0) x E1g(1) ...... B(1) | E2(1) x E1g(2) ...           x = halo exchange (one per pair)
//     main:  I(0) ........ | I(1) ................      | I(2) ...
// ONE exchange per pair (depth-2 halo, LocalPlan::ghost_rows): X^k of every remote row my perimeter rows read AND of the rows those
// read travels; E1g evaluates my perimeter rows and, redundantly, those ghost rows of X^(k+1) (same arithmetic as their owner,
// bit for bit), so E2 finds its remote operands in M without a second exchange.
// Buffers: A = complete input, Bf = output, M = perimeter + ring (+ ghost rows) of the intermediate field; A and Bf swap per pair.
void Smoother::relax_pairs_pipelined(uint64_t npairs, bool want_partials_last) {
    ensure_side();
    // Ordering between the two queues INSIDE the loop.  hipEventRecord / hipStreamWaitEvent are barrier packets, and every one of
    // them stalls its queue for 7-12 us on this part (rocprofv3 traces under profiles/): 10 % of a 4096^2 pass on the handle's
    // stream, a fifth of the chain of a 2048^2 pass.  So a dependency is a one-wave kernel on either side instead: the producer queue
    // runs k_queue_signal behind the producing kernel (bumps a counter in device memory), the consumer queue runs k_queue_wait in
    // front of the consuming one (spins, with sleeps, until the counter has reached its target; kernel boundaries inside a queue
    // cost well under a microsecond).  A waiter's producer is always enqueued before it, in host order, so the pair cannot
    // deadlock even if both streams share a hardware queue; a wait that is not met within tens of seconds raises sync_flags[2] and
    // the pass fails with TM_E_HIP instead of hanging the device.  Several multi-rank handles in ONE process (the virtual-rank tests) could
    // block each other through shared hardware queues: they use events.
    const bool use_flags = use_counters();
    uint32_t* border_done = sync_flags;
    uint32_t* inside_done = sync_flags + 1;
    uint32_t* sync_err = sync_flags + 2;
    uint32_t n_border = 0, n_inside = 0;   // signals sent so far
    if (use_flags) HIPCHK(hipMemsetAsync(sync_flags, 0, sizeof(uint32_t) * 4, stream));
    auto signal_border = [&]() {   // on side, behind the border pass
        if (use_flags) HIPCHK(launch_queue_signal(border_done, side));
        else HIPCHK(hipEventRecord(ev_to_main, side));
        n_border += 1;
    };
    auto wait_border = [&]() {     // on the handle's stream, in front of the interior pass
        if (use_flags) HIPCHK(launch_queue_wait(border_done, n_border, sync_err, stream));
        else HIPCHK(hipStreamWaitEvent(stream, ev_to_main, 0));
    };
    auto signal_inside = [&]() {   // on the handle's stream, behind the interior pass
        if (use_flags) HIPCHK(launch_queue_signal(inside_done, stream));
        else HIPCHK(hipEventRecord(ev_inside[n_inside & 1], stream));
        n_inside += 1;
    };
    auto wait_inside = [&](uint32_t upto) {   // on side, in front of the border pass: interior passes 0 .. upto-1 are done
        if (use_flags) HIPCHK(launch_queue_wait(inside_done, upto, sync_err, side));
        else HIPCHK(hipStreamWaitEvent(side, ev_inside[(upto - 1) & 1], 0));
    };
    const EdgeRowsDev& e1 = edge_nf_g.nrows ? edge_nf_g : edge_nf;
    auto edge_on_side = [&](const EdgeRowsDev& e, const double2* in, double2* out, int dot, uint32_t* signal = nullptr) {
        HIPCHK(launch_edge_rows(e, in, in, PQ, nullptr, out, opt.omega, MODE_RELAX, dot, partials + static_cast<size_t>(poff2_edge) * MAX_PARTIALS, side, signal));
    };
    auto exchange_on_side = [&](double2* vec) {
        exchange(vec, side);
        exchange_finish(side);
    };
    // With counters the two dependencies of the chain cost no launch of their own (at 2048^2 and below a pair is bound by the HOST:
    // ~8 launches plus the transport's calls per pair; rocprofv3 timeline in profiles/): the border pass spins on "interior pass k
    // is done" in its own prologue (a few dozen workgroups), and "border pass k is done" is published by the first thread of the
    // perimeter-row launch that follows it in the queue -- the interior pass k+1 that waits for it reads nothing that launch
    // writes (it touches no row within four of a side whose perimeter rows move) and writes nothing it reads.
    auto border_pass = [&](uint32_t inside_upto, int dot) {   // border workgroups; inside_upto > 0: after interior passes 0 .. inside_upto-1
        if (use_flags && inside_upto > 0) {
            QueueWait w;
            w.counter = inside_done;
            w.target = inside_upto;
            w.error = sync_err;
            relax2_launch(R2_BORDER, true, dot, side, &w);
        } else {
            if (inside_upto > 0) wait_inside(inside_upto);
            relax2_launch(R2_BORDER, true, dot, side);
        }
    };
    // K2x2 parts read X (input) and write U (output) through relax2_launch: keep X/U pointing at the pair in flight
    int dot = (npairs == 1 && want_partials_last) ? DOT_DELTA : DOT_NONE;
    fence(stream, side, ev_to_side);   // X is complete on the main stream
    warm_transport();
    exchange_on_side(X);
    edge_on_side(e1, X, M, DOT_NONE);    // E1g(0): perimeter rows (own + ghost) of the intermediate field
    relax2_launch(R2_INSIDE, false, dot);
    if (!use_flags || npairs == 1) signal_inside();   // with counters the announcement travels with the next wait (one launch)
    border_pass(0, dot);
    for (uint64_t k = 0; k < npairs; ++k) {
        // E2(k): perimeter rows of the pair's output (ghost operands: M's ghost rows); with counters it also announces border pass k
        const bool announce = k + 1 < npairs;
        if (use_flags && announce) {
            edge_on_side(edge_nf, M, U, dot, border_done);
            n_border += 1;
        } else {
            edge_on_side(edge_nf, M, U, dot);
            if (announce) signal_border();
        }
        std::swap(X, U);
        if (k + 1 == npairs) {
            fence(side, stream, ev_to_main);          // the main stream continues behind the whole chain
            break;
        }
        // pair k+1: input = this pair's output
        dot = (k + 2 == npairs && want_partials_last) ? DOT_DELTA : DOT_NONE;
        // interior pass k+1 reads rows the border of pair k wrote, and overwrites its input
        if (use_flags) {   // "interior pass k is done" + "wait for border k" in one launch; the last interior pass is announced on its own
            HIPCHK(launch_queue_signal_wait(inside_done, border_done, n_border, sync_err, stream));
            n_inside += 1;
        } else {
            wait_border();
        }
        relax2_launch(R2_INSIDE, false, dot);
        if (!use_flags) signal_inside();
        exchange_on_side(X);
        edge_on_side(e1, X, M, DOT_NONE);             // E1g(k+1)
        border_pass(static_cast<uint32_t>(k + 1), dot);   // border k+1 reads rows interior pass k wrote, and overwrites ITS input
    }
    if (use_flags) {
        HIPCHK(hipMemcpyAsync(h_flags, sync_flags, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, stream));
        flags_pending = true;
    }
}

void Smoother::relax_sweeps(uint64_t n, tm_stats& st) {
    int last_nwg = nwg_apply;
    uint64_t k = 0;
    if ((fuse_triples || triples_coupled) && n >= 3) {   // n = 3 a + 2 b + c, as many triples as possible (n = 4: one triple + one single sweep)
        const uint64_t ntriples = n / 3;
        if (fuse_triples) {
            for (uint64_t q = 0; q < ntriples; ++q) relax_triple(q + 1 == ntriples && n % 3 == 0);
            last_nwg = nwg_apply3;
        } else {
            relax_triples_coupled(ntriples, n % 3 == 0);
            last_nwg = nwg_apply3 + (levels_fused ? fused_levels.nstrips : edge_L[2].nwg);
        }
        k = 3 * ntriples;
        st.operator_sweeps += k;
        outer_done += k;
    }
    if (fuse_pairs && n - k >= 2) {
        const uint64_t npairs = (n - k) / 2;
        const bool last_is_pair = (n - k) % 2 == 0;
        if ((has_hooks && (n_send > 0 || n_ghost > 0)) || pipelined_single) {
            relax_pairs_pipelined(npairs, last_is_pair);
        } else {
            for (uint64_t q = 0; q < npairs; ++q) relax_pair(q + 1 == npairs && last_is_pair);
        }
        k += 2 * npairs;
        st.operator_sweeps += 2 * npairs;
        outer_done += 2 * npairs;
        last_nwg = nwg_apply2;
    }
    for (; k < n; ++k) {
        last_nwg = nwg_apply;
        if (white && outer_done > 0) white_launch(1);
        // one fused sweep: U = X + omega D^-1 (b - A(X) X), partial sums of (U - X)^2
        apply(X, U, MODE_RELAX, DOT_DELTA, nullptr, X, opt.omega);
        std::swap(X, U);
        st.operator_sweeps += 1;
        outer_done += 1;
    }
    if (profile && prof_open) {   // a group bracket the call leaves open: close it behind its last launch; the next call starts a new one
        profile_close(stream);
        prof_phase = 0;
    }
    if (n) {
        reduce(last_nwg);   // partial sums of the LAST sweep -> sum (x_old - x_new)^2, sum (y_old - y_new)^2
        HIPCHK(hipMemcpyAsync(h_red, red, sizeof(double) * MAX_PARTIALS, hipMemcpyDeviceToHost, stream));
        sync();
        if (flags_pending) {
            flags_pending = false;
            const bool timed_out = h_flags[2] != 0 || h_flags[6] != 0;
            h_flags[2] = h_flags[6] = 0;
            if (timed_out) throw TmError(TM_E_HIP, "a device-side dependency between the interior pass and the chain of a sweep pair / triple was not met within its time limit");
        }
        st.last_dx2 = h_red[0];
        st.last_dy2 = h_red[1];
        st.last_residual = (h_red[0] + h_red[1]) * (h_red[0] + h_red[1]);
        // for a Jacobi sweep the displacement is omega times the scaled residual
        st.scaled_residual_rms = std::sqrt((h_red[0] + h_red[1]) / (2.0 * static_cast<double>(dof_global))) / opt.omega;
    }
}

void Smoother::iterate(uint64_t iterations, tm_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    tm_stats st;
    std::memset(&st, 0, sizeof(st));
    const tm_log_fn sink = g_log_sink;
    if (opt.inner == TM_INNER_RELAX && !sink) {
        relax_sweeps(iterations, st);
        st.outer_iterations = iterations;
    } else {
        // smooth.zig:104-137: "iteration: n" before the fill, "\tresidual: ..." after the solve.  A relaxation sweep IS an
        // outer iteration, so a sink makes every sweep end with its own reduction (one sweep per pass, see tm_set_log).
        for (uint64_t n = 0; n < iterations; ++n) {
            if (sink) sink(g_log_ctx, 0, n, 0.0);
            if (opt.inner == TM_INNER_RELAX) relax_sweeps(1, st);
            else st.not_converged += picard_solve(st);
            st.outer_iterations += 1;
            if (sink) sink(g_log_ctx, 1, n, st.last_residual);
        }
    }
    sync();
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
}

// Outer iterations until the UPDATE of the last one, sqrt((sum dx^2 + sum dy^2) / dof) over all nodes of the mesh, is <= tol: the
// quantity the reference itself forms and logs every iteration (smooth.zig:112-137, it prints its square squared) -- a distance
// between consecutive iterates, i.e. a statement about convergence of the coordinates, which the scaled residual (~ h^2 x the
// displacement a point-Jacobi step would make) is not on a fine mesh.  Relax mode tests every 32 sweeps.
bool Smoother::iterate_until_update(uint64_t max_iterations, double tol, tm_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    tm_stats st;
    std::memset(&st, 0, sizeof(st));
    bool reached = false;
    const tm_log_fn sink = g_log_sink;
    while (st.outer_iterations < max_iterations) {
        if (opt.inner == TM_INNER_RELAX) {
            const uint64_t n = std::min<uint64_t>(32, max_iterations - st.outer_iterations);
            relax_sweeps(n, st);
            st.outer_iterations += n;
        } else {
            if (sink) sink(g_log_ctx, 0, st.outer_iterations, 0.0);
            st.not_converged += picard_solve(st);
            if (sink) sink(g_log_ctx, 1, st.outer_iterations, st.last_residual);
            st.outer_iterations += 1;
        }
        if (std::sqrt((st.last_dx2 + st.last_dy2) / static_cast<double>(dof_global)) <= tol) {
            reached = true;
            break;
        }
    }
    sync();
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return reached;
}

// Outer iterations until the scaled nonlinear residual sqrt(||D^-1 (b - A(X) X)||^2 / (2 dof)) is <= tol (the reference has no
// stop test; its iteration count comes from the input file).  Picard modes test the residual every iteration (it
// is the inner solve's start residual, so the test costs nothing); relax mode every 32 sweeps.
bool Smoother::iterate_until(uint64_t max_iterations, double tol, tm_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    tm_stats st;
    std::memset(&st, 0, sizeof(st));
    bool reached = false;
    if (opt.inner == TM_INNER_RELAX) {
        while (st.outer_iterations < max_iterations) {
            const uint64_t n = std::min<uint64_t>(32, max_iterations - st.outer_iterations);
            relax_sweeps(n, st);
            st.outer_iterations += n;
            if (st.scaled_residual_rms <= tol) {
                reached = true;
                break;
            }
        }
    } else {
        stop_tol = tol;
        try {
            while (true) {
                const tm_log_fn sink = g_log_sink;
                if (sink) sink(g_log_ctx, 0, st.outer_iterations, 0.0);
                const int rc = picard_solve(st);
                if (rc == 2) {
                    reached = true;
                    break;
                }
                if (sink) sink(g_log_ctx, 1, st.outer_iterations, st.last_residual);
                st.not_converged += rc;
                st.outer_iterations += 1;
                if (st.outer_iterations >= max_iterations) break;
            }
        } catch (...) {
            stop_tol = 0.0;
            throw;
        }
        stop_tol = 0.0;
    }
    sync();
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return reached;
}

void Smoother::profile_read(double* ms_total, uint64_t* launches, uint64_t* timed) {
    sync();
    double total = 0.0;
    for (size_t k = 0; k < ev_used; ++k) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev_start[k], ev_stop[k]));
        total += ms;
    }
    if (ms_total) *ms_total = total;
    if (launches) *launches = prof_launches;
    if (timed) *timed = prof_timed;
    ev_used = 0;
    prof_launches = 0;
    prof_timed = 0;
    prof_phase = 0;
}

// ------------------------------------------------------------------ introspection
void Smoother::ensure_tmp() {
    if (tmpA) return;
    if (r && v) {   // BiCGStab workspaces are free between iterate() calls
        tmpA = r;
        tmpB = v;
        return;
    }
    tmpA = arena.alloc_n<double2>(static_cast<uint64_t>(n_local));
    tmpB = arena.alloc_n<double2>(static_cast<uint64_t>(n_local));
}

void Smoother::apply_host(const double* in_xy, double* out_xy, int scaled) {
    if (has_hooks) throw TmError(TM_E_UNSUPPORTED, "tm_smoother_apply is available on single-process handles only");
    ensure_tmp();
    HIPCHK(hipMemcpyAsync(tmpA, in_xy, sizeof(double2) * n_owned, hipMemcpyHostToDevice, stream));
    apply(tmpA, tmpB, scaled ? MODE_SCALED : MODE_RAW, DOT_NONE, nullptr, X, 0.0);
    HIPCHK(hipMemcpyAsync(out_xy, tmpB, sizeof(double2) * n_owned, hipMemcpyDeviceToHost, stream));
    sync();
}

// ---- the reference's assembled system on the device (introspection).  Pattern: RowCompressedMatrixSystem2d.init (smooth.zig:309-385) --
// interior rows nine columns in ascending id, perimeter rows the plan's columns (ascending global id = the reference's order); values:
// system.fill for the CURRENT coordinates (smooth.zig:923-1113) by k_assemble_interior / k_assemble_edge, x and y systems (they differ in
// the sliding rows, smooth.zig:1115-1165).
void Smoother::csr_build_pattern() {
    if (has_hooks) throw TmError(TM_E_UNSUPPORTED, "the assembled system is available on single-process handles only");
    if (!csr.h_p.empty()) return;
    const int64_t n = n_owned;
    std::vector<int32_t> count(static_cast<size_t>(n), 9);
    std::vector<const PlanRow*> perim(static_cast<size_t>(n), nullptr);
    for (const PlanRow& pr : lp.rows) {
        const int64_t l = lp.to_local(pr.gid);
        if (l < 0 || l >= n) throw TmError(TM_E_TOPOLOGY, "internal: perimeter row outside the owned rows");
        count[l] = pr.ncols;
        perim[l] = &pr;
    }
    uint64_t nnz = 0;
    for (int64_t r = 0; r < n; ++r) nnz += static_cast<uint64_t>(count[r]);
    if (nnz >= (uint64_t{1} << 31)) throw TmError(TM_E_SIZE, "the assembled system has more than 2^31 non-zeros");
    csr.h_p.resize(static_cast<size_t>(n) + 1);
    csr.h_i.resize(nnz);
    int32_t at = 0;
    for (size_t kb = 0; kb < lp.owned_blocks.size(); ++kb) {
        const int64_t b = lp.owned_blocks[kb], ls = lp.local_start[kb], bj = topo.nj[b];
        const int64_t nb = topo.ni[b] * bj;
        for (int64_t f = 0; f < nb; ++f) {
            const int64_t r = ls + f;
            csr.h_p[r] = at;
            if (const PlanRow* pr = perim[r]) {
                for (int q = 0; q < pr->ncols; ++q) csr.h_i[at++] = static_cast<int32_t>(lp.to_local(pr->col[q]));
            } else {
                for (int64_t di = -1; di <= 1; ++di)
                    for (int64_t dj = -1; dj <= 1; ++dj) csr.h_i[at++] = static_cast<int32_t>(r + di * bj + dj);
            }
        }
    }
    csr.h_p[n] = at;
    csr.nnz = nnz;
}

void Smoother::csr_release() {
    for (void* q : {static_cast<void*>(csr.p), static_cast<void*>(csr.i), static_cast<void*>(csr.vx), static_cast<void*>(csr.vy)})
        if (q) (void)hipFree(q);
    csr.p = csr.i = nullptr;
    csr.vx = csr.vy = nullptr;
}

void Smoother::csr_fill_values() {
    csr_build_pattern();
    if (!csr.p) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&csr.p), sizeof(int32_t) * csr.h_p.size()));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&csr.i), sizeof(int32_t) * std::max<size_t>(1, csr.h_i.size())));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&csr.vx), sizeof(double) * std::max<uint64_t>(1, csr.nnz)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&csr.vy), sizeof(double) * std::max<uint64_t>(1, csr.nnz)));
        HIPCHK(hipMemcpyAsync(csr.p, csr.h_p.data(), sizeof(int32_t) * csr.h_p.size(), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(csr.i, csr.h_i.data(), sizeof(int32_t) * csr.h_i.size(), hipMemcpyHostToDevice, stream));
    }
    for (size_t kb = 0; kb < lp.owned_blocks.size(); ++kb) {
        const int64_t b = lp.owned_blocks[kb], ls = lp.local_start[kb];
        HIPCHK(launch_assemble_interior(X + ls, PQ ? PQ + ls : nullptr, static_cast<int>(topo.ni[b]), static_cast<int>(topo.nj[b]), csr.p + ls, csr.vx, csr.vy, stream));
    }
    HIPCHK(launch_assemble_edge(edge, X, PQ, csr.p, csr.vx, csr.vy, stream));
}

uint64_t Smoother::assemble_csr_host(int32_t* Ap, int32_t* Ai, double* Ax_x, double* Ax_y, uint64_t nnz_capacity) {
    csr_build_pattern();
    if (!Ap && !Ai && !Ax_x && !Ax_y) return csr.nnz;   // size query
    if (Ap) std::memcpy(Ap, csr.h_p.data(), sizeof(int32_t) * csr.h_p.size());
    if ((Ai || Ax_x || Ax_y) && nnz_capacity < csr.nnz) throw TmError(TM_E_SIZE, "nnz_capacity is smaller than the system's non-zero count");
    if (Ai) std::memcpy(Ai, csr.h_i.data(), sizeof(int32_t) * csr.h_i.size());
    if (Ax_x || Ax_y) {
        csr_fill_values();
        if (Ax_x) HIPCHK(hipMemcpyAsync(Ax_x, csr.vx, sizeof(double) * csr.nnz, hipMemcpyDeviceToHost, stream));
        if (Ax_y) HIPCHK(hipMemcpyAsync(Ax_y, csr.vy, sizeof(double) * csr.nnz, hipMemcpyDeviceToHost, stream));
        sync();
        csr_release();
    }
    return csr.nnz;
}

void Smoother::apply_reference_host(const double* in_xy, double* out_xy) {
    ensure_tmp();
    csr_fill_values();
    HIPCHK(hipMemcpyAsync(tmpA, in_xy, sizeof(double2) * n_owned, hipMemcpyHostToDevice, stream));
    HIPCHK(launch_csr_product(n_owned, csr.p, csr.i, csr.vx, csr.vy, tmpA, tmpB, stream));
    HIPCHK(hipMemcpyAsync(out_xy, tmpB, sizeof(double2) * n_owned, hipMemcpyDeviceToHost, stream));
    sync();
    csr_release();
}

void Smoother::rhs_host(double* rhs_xy) {
    if (has_hooks) throw TmError(TM_E_UNSUPPORTED, "tm_smoother_rhs is available on single-process handles only");
    ensure_tmp();
    HIPCHK(hipMemsetAsync(tmpA, 0, sizeof(double2) * n_local, stream));
    HIPCHK(launch_edge_rhs(edge, X, PQ, tmpA, 0, nullptr, stream));
    HIPCHK(hipMemcpyAsync(rhs_xy, tmpA, sizeof(double2) * n_owned, hipMemcpyDeviceToHost, stream));
    sync();
}

void Smoother::control_function_host(double* pq) {
    if (!PQ) {
        std::memset(pq, 0, sizeof(double) * 2 * n_owned);
        return;
    }
    HIPCHK(hipMemcpyAsync(pq, PQ, sizeof(double2) * n_owned, hipMemcpyDeviceToHost, stream));
    sync();
}

// One owned block as planes (x, y and optionally P, Q), element j*ni + i: what cgns.zig:75-154 hands to cg_coord_write /
// cg_field_write, transposed on the device.
void Smoother::export_soa_host(int64_t block, double* x, double* y, double* p, double* q) {
    const auto it = std::lower_bound(lp.owned_blocks.begin(), lp.owned_blocks.end(), block);
    if (block < 0 || it == lp.owned_blocks.end() || *it != block) throw TmError(TM_E_ARG, "block is not owned by this rank");
    const int64_t ls = lp.local_start[it - lp.owned_blocks.begin()];
    const int bi = static_cast<int>(topo.ni[block]), bj = static_cast<int>(topo.nj[block]);
    const size_t n = static_cast<size_t>(bi) * bj;
    // own scratch (two planes of the largest block exported so far): on a relax handle every field buffer, U included, carries the
    // `fixed` boundary coordinates on its perimeter between iterate() calls (prefill_fixed) -- none of them is free to scribble on
    if (export_bytes < sizeof(double) * 2 * n) {
        if (export_buf) (void)hipFree(export_buf);
        export_buf = nullptr;
        export_bytes = 0;
        if (hipMalloc(&export_buf, sizeof(double) * 2 * n) != hipSuccess) throw TmError(TM_E_MEMORY, "hipMalloc failed (export planes)");
        export_bytes = sizeof(double) * 2 * n;
    }
    double* planes = static_cast<double*>(export_buf);
    HIPCHK(launch_soa_planes(X + ls, planes, planes + n, bi, bj, stream));
    HIPCHK(hipMemcpyAsync(x, planes, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(y, planes + n, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    if (p && q) {
        if (PQ) {   // same scratch, after the copies above in stream order
            HIPCHK(launch_soa_planes(PQ + ls, planes, planes + n, bi, bj, stream));
            HIPCHK(hipMemcpyAsync(p, planes, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(q, planes + n, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
        } else {   // laplace: the control function is identically zero (wall_control_function.zig:22-54)
            std::memset(p, 0, sizeof(double) * n);
            std::memset(q, 0, sizeof(double) * n);
        }
    }
    sync();
}

}  // namespace tmh
