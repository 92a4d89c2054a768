// Built-in transport behind tm_comm_hooks: RCCL point-to-point (over xGMI) and all-reduce, issued by the library itself.
//
// The halo exchange of a sweep is two small messages per neighbour (interface row + first-interior row, 2 x 64 KiB at
// 4096 columns).  Issued from a host-language callback that costs ~50-70 us per exchange (measured with
// torch.distributed's batch_isend_irecv, tools/ubench/p2p_host_cost.py), which is more than half of what the GPU needs for
// the two sweeps in between; issued from here it is a grouped ncclSend/ncclRecv on a side stream plus two events.
//
// librccl is loaded at run time (dlopen) so that the process uses ONE RCCL -- the copy the host framework already loaded,
// when there is one (the caller passes its path) -- and so that libtm_hip.so has no link-time dependency on it.
#include "tm_api_util.hpp"
#include "tm_kernels.h"
#include "tm_plan.hpp"
#include "tm_smoother.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace tmh {

// ---- the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes; ncclFloat64 = 8, ncclSum = 0)
struct NcclUniqueId {
    char internal[128];
};
typedef struct ncclComm* NcclComm;
enum { NCCL_SUCCESS = 0, NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static RcclApi& rccl_api(const char* path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return g_rccl;
    const char* candidates[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::string tried;
    for (const char* c : candidates) {
        if (!c || !*c) continue;
        g_rccl.handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
        tried += std::string(c) + ": " + dlerror() + "; ";
    }
    if (!g_rccl.handle) throw TmError(TM_E_COMM, "cannot load librccl (" + tried + ")");
    auto sym = [&](const char* name) {
        void* p = dlsym(g_rccl.handle, name);
        if (!p) throw TmError(TM_E_COMM, std::string("librccl lacks ") + name);
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    return g_rccl;
}

static void nccl_check(int rc, const char* what) {
    if (rc != NCCL_SUCCESS) throw TmError(TM_E_COMM, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

}  // namespace tmh

using namespace tmh;

// One communicator of the job + the side stream its transfers run on.
struct tm_rccl_comm {
    NcclComm comm = nullptr;
    int32_t rank = 0, nranks = 1;
    // Where the transfers are enqueued.  Default: on the stream the handle names for the call -- the chain of a sweep pair
    // (perimeter rows -> exchange -> perimeter rows -> border workgroups) is bound by dependency latency, and a hop to another
    // queue and back costs two event waits of ~13 us each on this part.  TM_RCCL_OWN_STREAM=1 selects the other form: an own
    // high-priority stream fenced with two events, which lets a Krylov-path exchange run beside the interior-row kernel.
    bool own_stream = false;
    bool allow_triples = true;   // the depth the tables below were built for (tm_rccl_hooks: by topology; tm_rccl_hooks_for: by the solver options too)
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    // exchange pattern of the partition given to tm_rccl_hooks (rows are double2)
    std::vector<int32_t> owner, peer;
    std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;
};

namespace tmh {

// What one rank hands to ncclSend / ncclRecv per exchange: for every neighbouring rank an offset + count (rows) into the send
// buffer the handle passes (its own vector when every send list is one run of local rows, else the packed rows) and into the
// ghost segment.  A pure function of topology + partition, so that the property that matters -- rank a's send to b has exactly the
// size of b's receive from a, for every pair, or ncclGroupEnd never returns -- is checkable without a communicator
// (tm_rccl_peer_table_build, tests/test_rccl_peer_tables.py).
struct PeerTable {
    std::vector<int32_t> peer;
    std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;
    int64_t send_rows = 0, recv_rows = 0;   // extent of the two buffers the offsets index
    bool direct_send = false;
};

static PeerTable peer_table_of(const LocalPlan& lp) {
    PeerTable t;
    t.peer.assign(lp.peer_rank.begin(), lp.peer_rank.end());
    t.direct_send = lp.direct_send;
    if (lp.direct_send) t.send_off.assign(lp.send_first.begin(), lp.send_first.end());   // the handle passes the vector itself as send buffer
    else t.send_off.assign(lp.send_off.begin(), lp.send_off.end());
    t.send_cnt.assign(lp.send_cnt.begin(), lp.send_cnt.end());
    t.recv_off.assign(lp.recv_off.begin(), lp.recv_off.end());
    t.recv_cnt.assign(lp.recv_cnt.begin(), lp.recv_cnt.end());
    t.send_rows = lp.direct_send ? lp.n_owned + static_cast<int64_t>(lp.ghost_gid.size()) : static_cast<int64_t>(lp.send_ids.size());
    t.recv_rows = static_cast<int64_t>(lp.ghost_gid.size());
    return t;
}

}  // namespace tmh

namespace {

int rccl_exchange(void* ctx, const double* send_buf, double* recv_buf, void* stream) {
    tm_rccl_comm* c = static_cast<tm_rccl_comm*>(ctx);
    try {
        if (c->peer.empty()) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        hipStream_t xs = c->own_stream ? c->stream : s;
        if (c->own_stream) {
            HIPCHK(hipEventRecord(c->ready, s));   // the pack kernel has filled send_buf
            HIPCHK(hipStreamWaitEvent(c->stream, c->ready, 0));
        }
        nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
        for (size_t k = 0; k < c->peer.size(); ++k) {
            if (c->recv_cnt[k])
                nccl_check(g_rccl.Recv(recv_buf + 2 * c->recv_off[k], static_cast<size_t>(2 * c->recv_cnt[k]), NCCL_FLOAT64, c->peer[k], c->comm, xs), "ncclRecv");
            if (c->send_cnt[k])
                nccl_check(g_rccl.Send(send_buf + 2 * c->send_off[k], static_cast<size_t>(2 * c->send_cnt[k]), NCCL_FLOAT64, c->peer[k], c->comm, xs), "ncclSend");
        }
        nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
        if (c->own_stream) HIPCHK(hipEventRecord(c->done, c->stream));
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 1;
    }
}

int rccl_exchange_wait(void* ctx, void* stream) {
    tm_rccl_comm* c = static_cast<tm_rccl_comm*>(ctx);
    if (c->peer.empty() || !c->own_stream) return 0;   // enqueued on `stream` itself: already ordered
    return hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done, 0) == hipSuccess ? 0 : 1;
}

int rccl_allreduce(void* ctx, double* buf, int32_t n, void* stream) {
    tm_rccl_comm* c = static_cast<tm_rccl_comm*>(ctx);
    try {
        if (c->nranks == 1) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (!c->own_stream) {
            nccl_check(g_rccl.AllReduce(buf, buf, static_cast<size_t>(n), NCCL_FLOAT64, NCCL_SUM, c->comm, s), "ncclAllReduce");
            return 0;
        }
        // on the transfer stream too: operations of one communicator stay in one queue
        HIPCHK(hipEventRecord(c->ready, s));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ready, 0));
        nccl_check(g_rccl.AllReduce(buf, buf, static_cast<size_t>(n), NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream), "ncclAllReduce");
        HIPCHK(hipEventRecord(c->done, c->stream));
        HIPCHK(hipStreamWaitEvent(s, c->done, 0));
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 1;
    }
}

}  // namespace

namespace tmh {
int rccl_hooks_allow_triples(const tm_comm_hooks* h) {
    if (!h || h->exchange != rccl_exchange || !h->ctx) return -1;
    return static_cast<const tm_rccl_comm*>(h->ctx)->allow_triples ? 1 : 0;
}
}  // namespace tmh

extern "C" {

int tm_rccl_unique_id(const char* librccl_path, void* id_out) {
    return guarded([&]() {
        if (!id_out) throw TmError(TM_E_ARG, "null argument");
        RcclApi& api = rccl_api(librccl_path);
        NcclUniqueId id;
        nccl_check(api.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id_out, &id, sizeof(id));
        return TM_OK;
    });
}

int tm_rccl_comm_create(const char* librccl_path, const void* id, int32_t rank, int32_t nranks, tm_rccl_comm** out) {
    return guarded([&]() {
        if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) throw TmError(TM_E_ARG, "bad argument");
        RcclApi& api = rccl_api(librccl_path);
        auto c = std::make_unique<tm_rccl_comm>();
        c->rank = rank;
        c->nranks = nranks;
        NcclUniqueId uid;
        std::memcpy(&uid, id, sizeof(uid));
        nccl_check(api.CommInitRank(&c->comm, nranks, uid, rank), "ncclCommInitRank");
        const char* own = std::getenv("TM_RCCL_OWN_STREAM");
        c->own_stream = own && own[0] == '1';
        int least = 0, greatest = 0;   // transfers are tiny and latency-critical
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
        HIPCHK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
        *out = c.release();
        return TM_OK;
    });
}

void tm_rccl_comm_destroy(tm_rccl_comm* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#ifdef TM_DEBUG_EXPORTS   // `make DEBUG_EXPORTS=1`: measurement helpers, not part of include/tm_hip.h (tools/README.md)
// internal (tools/split_path_cost.py): hooks that move nothing, to time the multi-rank schedule of ONE rank without its peers.
// The ghost rows stay stale, so the coordinates are meaningless.  Not part of the drop-in surface.
int tm_debug_null_hooks(int32_t rank, int32_t nranks, const int32_t* owner, tm_comm_hooks* hooks) {
    std::memset(hooks, 0, sizeof(*hooks));
    hooks->rank = rank;
    hooks->nranks = nranks;
    hooks->owner = owner;
    // TM_NULL_EXCHANGE_US=t: every exchange occupies the stream it is issued on for t microseconds (a one-wave kernel): the device time
    // of a real transfer ON the chain, without peers
    static double delay_us = 0.0;
    if (const char* e = std::getenv("TM_NULL_EXCHANGE_US")) delay_us = std::atof(e);
    hooks->exchange = [](void*, const double*, double*, void* stream) { return launch_delay_us(delay_us, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : 1; };
    hooks->exchange_wait = [](void*, void*) { return 0; };
    hooks->allreduce_sum = [](void*, double*, int32_t, void*) { return 0; };
    return TM_OK;
}

// internal (tools/ubench/rccl_selftest.py): device-side time of one grouped exchange of 2 x `rows` double2 with the own rank as
// peer -- launch + handshake cost of the RCCL point-to-point path on this GPU (the wire is not exercised).
int tm_debug_rccl_selftest(tm_rccl_comm* c, int64_t rows, int32_t iters, double* us_per_exchange) {
    return guarded([&]() {
        if (!c || rows <= 0 || iters <= 0 || !us_per_exchange) throw TmError(TM_E_ARG, "bad argument");
        double *snd = nullptr, *rcv = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&snd), sizeof(double) * 2 * rows * 2));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&rcv), sizeof(double) * 2 * rows * 2));
        HIPCHK(hipMemset(snd, 0, sizeof(double) * 2 * rows * 2));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        auto once = [&]() {
            nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
            for (int k = 0; k < 2; ++k) {   // two "neighbours", like a middle rank of a strip
                nccl_check(g_rccl.Recv(rcv + 2 * rows * k, static_cast<size_t>(2 * rows), NCCL_FLOAT64, c->rank, c->comm, c->stream), "ncclRecv");
                nccl_check(g_rccl.Send(snd + 2 * rows * k, static_cast<size_t>(2 * rows), NCCL_FLOAT64, c->rank, c->comm, c->stream), "ncclSend");
            }
            nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
        };
        for (int i = 0; i < 10; ++i) once();
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipEventRecord(e0, c->stream));
        for (int i = 0; i < iters; ++i) once();
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *us_per_exchange = 1e3 * ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(snd);
        (void)hipFree(rcv);
        return TM_OK;
    });
}
#endif   // TM_DEBUG_EXPORTS

int tm_rccl_peer_table_build(const tm_mesh_desc* mesh, const int32_t* owner, int32_t rank, int32_t nranks, tm_rccl_peer_table* out) {
    return guarded([&]() {
        if (!mesh || !owner || !out || nranks < 1 || rank < 0 || rank >= nranks) throw TmError(TM_E_ARG, "bad argument");
        std::memset(out, 0, sizeof(*out));
        const Topology topo = topo_of(mesh);
        std::vector<int32_t> own(owner, owner + topo.nblocks());
        for (int32_t o : own)
            if (o < 0 || o >= nranks) throw TmError(TM_E_ARG, "owner rank out of range");
        const PeerTable t = peer_table_of(build_local_plan(topo, build_rows(topo), own, rank, nranks));
        const size_t n = t.peer.size();
        out->npeers = static_cast<int32_t>(n);
        out->direct_send = t.direct_send ? 1 : 0;
        out->send_rows = t.send_rows;
        out->recv_rows = t.recv_rows;
        auto dup = [&](const auto& v) {
            using T = typename std::decay_t<decltype(v)>::value_type;
            T* p = static_cast<T*>(std::malloc(sizeof(T) * (n ? n : 1)));
            if (!p) throw TmError(TM_E_MEMORY, "malloc failed");
            if (n) std::memcpy(p, v.data(), sizeof(T) * n);
            return p;
        };
        out->peer = dup(t.peer);
        out->send_off = dup(t.send_off);
        out->send_cnt = dup(t.send_cnt);
        out->recv_off = dup(t.recv_off);
        out->recv_cnt = dup(t.recv_cnt);
        return TM_OK;
    });
}

void tm_rccl_peer_table_free(tm_rccl_peer_table* t) {
    if (!t) return;
    std::free(t->peer);
    std::free(t->send_off);
    std::free(t->send_cnt);
    std::free(t->recv_off);
    std::free(t->recv_cnt);
    std::memset(t, 0, sizeof(*t));
}

static int rccl_hooks_impl(tm_rccl_comm* c, const tm_mesh_desc* mesh, const int32_t* owner, bool allow_triples, tm_comm_hooks* hooks);

int tm_rccl_hooks(tm_rccl_comm* c, const tm_mesh_desc* mesh, const int32_t* owner, tm_comm_hooks* hooks) {
    return rccl_hooks_impl(c, mesh, owner, true, hooks);
}

// The same for a handle whose options are known: a handle that never runs sweep triples (Krylov modes, the White control function,
// TM_OPT_SINGLE_SWEEP) exchanges the depth-2 halo only -- a third fewer rows per exchange on blocks of 2^19 nodes and more.
int tm_rccl_hooks_for(tm_rccl_comm* c, const tm_mesh_desc* mesh, const int32_t* owner, const tm_solver_opt* opt, const tm_control_fn* cf, tm_comm_hooks* hooks) {
    if (!opt) return rccl_hooks_impl(c, mesh, owner, true, hooks);
    const tm_control_fn laplace{TM_CF_LAPLACE, 0, 0.0, 0.0};
    return rccl_hooks_impl(c, mesh, owner, triples_wanted(*opt, cf ? *cf : laplace), hooks);
}

static int rccl_hooks_impl(tm_rccl_comm* c, const tm_mesh_desc* mesh, const int32_t* owner, bool allow_triples, tm_comm_hooks* hooks) {
    return guarded([&]() {
        if (!c || !mesh || !owner || !hooks) throw TmError(TM_E_ARG, "null argument");
        c->allow_triples = allow_triples;
        const Topology topo = topo_of(mesh);
        c->owner.assign(owner, owner + topo.nblocks());
        for (int32_t o : c->owner)
            if (o < 0 || o >= c->nranks) throw TmError(TM_E_ARG, "owner rank out of range");
        const LocalPlan lp = build_local_plan(topo, build_rows(topo), c->owner, c->rank, c->nranks, allow_triples);
        PeerTable t = peer_table_of(lp);
        c->peer = std::move(t.peer);
        c->send_off = std::move(t.send_off);
        c->send_cnt = std::move(t.send_cnt);
        c->recv_off = std::move(t.recv_off);
        c->recv_cnt = std::move(t.recv_cnt);
        std::memset(hooks, 0, sizeof(*hooks));
        hooks->ctx = c;
        hooks->rank = c->rank;
        hooks->nranks = c->nranks;
        hooks->owner = c->owner.data();
        hooks->exchange = rccl_exchange;
        hooks->exchange_wait = rccl_exchange_wait;
        hooks->allreduce_sum = rccl_allreduce;
        return TM_OK;
    });
}

}  // extern "C"
