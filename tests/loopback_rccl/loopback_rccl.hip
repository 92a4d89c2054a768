// TEST INFRASTRUCTURE -- not part of the product, never loaded by turbomesh_amd/ (tests/test_capi_symbols.py greps for that).
//
// A stand-in for librccl that lets SEVERAL ranks share ONE GPU, so that libtm_hip's own transport (csrc/tm_rccl.cpp: grouped
// ncclRecv / ncclSend per neighbouring rank + ncclAllReduce, dlopen'ed by path) can run with N > 1 on a one-GPU box.  Real RCCL refuses
// two ranks on one device.  It exports exactly the symbols tm_rccl.cpp resolves and keeps their contract:
//   * ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd are enqueued on the hipStream_t they name; the send buffer is read after
//     everything enqueued on that stream before, the receive buffer is complete for everything enqueued after; the host never blocks on
//     the device (GroupEnd returns once the kernels are enqueued);
//   * sends and receives of a group proceed together (a rank may list its receive first: no deadlock);
//   * ncclAllReduce(sum, double) is in place, in rank order, so every rank reads the same bits.
// Mechanism: every rank owns a MAILBOX in device memory (hipMalloc), exported to the other processes with hipIpcGetMemHandle through a POSIX
// shared-memory segment named by the unique id.  A message from s to d is pushed by s's kernels into d's mailbox (slot ring per ordered pair,
// `posted` counter bumped with a release behind the copy), and pulled by d's kernels into the user's receive buffer (`consumed` counter bumped
// behind it: flow control for the ring).  The waits are one-wave kernels that poll the counters with a time limit on the 100 MHz constant
// clock; a wait that runs out raises a flag in pinned host memory and every later call returns ncclSystemError instead of hanging the box.
//
// build: make -C tests/loopback_rccl   (hipcc --offload-arch=gfx950 -shared -fPIC)
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr int RING = 4;                           // slots per ordered pair
constexpr uint32_t MAGIC = 0x746d6c62;            // "tmlb"
constexpr size_t REDUCE_MAX = 4096;               // doubles per all-reduce

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
enum { ncclFloat64 = 8, ncclSum = 0 };

struct UniqueId {   // the 128 opaque bytes of ncclUniqueId
    uint32_t magic;
    uint32_t pid;
    uint64_t nonce;
    char name[64];
    char pad[48];
};
static_assert(sizeof(UniqueId) == 128, "ncclUniqueId is 128 bytes");

struct RankSlot {
    std::atomic<int> ready;       // mailbox allocated, handle below valid
    std::atomic<int> departed;
    int pid;
    hipIpcMemHandle_t mailbox;
};
struct Shared {
    std::atomic<int> created;
    int nranks;
    uint64_t slot_bytes;
    RankSlot r[MAX_RANKS];
};

// Device layout of one rank's mailbox: counters first (one 64-byte line each, so that polls of different pairs never share a line), then the
// slot rings of the MAX_RANKS possible senders.
struct Layout {
    size_t slot_bytes;
    size_t posted(int src) const { return 64 * static_cast<size_t>(src); }                        // written by the sender `src`
    size_t consumed(int src) const { return 64 * static_cast<size_t>(MAX_RANKS + src); }          // written by the owner
    size_t slot(int src, uint32_t seq) const { return 64 * 2 * MAX_RANKS + (static_cast<size_t>(src) * RING + seq % RING) * slot_bytes; }
    size_t total() const { return 64 * 2 * MAX_RANKS + static_cast<size_t>(MAX_RANKS) * RING * slot_bytes; }
};

struct Op {
    bool send;
    char* buf;
    size_t bytes;
    int peer;
    struct Comm* comm;
    hipStream_t stream;
};

struct Comm {
    int rank = 0, nranks = 1;
    Shared* sh = nullptr;
    size_t sh_bytes = 0;
    std::string name;
    Layout lay{};
    char* mine = nullptr;                      // my mailbox
    char* box[MAX_RANKS] = {};                 // every rank's mailbox as mapped here (box[rank] == mine)
    uint32_t sent[MAX_RANKS] = {};             // chunks pushed to each destination so far (host counter = next sequence number)
    uint32_t rcvd[MAX_RANKS] = {};             // chunks pulled from each source so far
    uint32_t* err = nullptr;                   // pinned host word: a device-side wait ran into its limit
    double* red = nullptr;                     // scratch of the all-reduce: [nranks][REDUCE_MAX]
    long long limit_ticks = 0;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

// ---------------------------------------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(64) void lb_wait(const uint32_t* counter, uint32_t target, uint32_t* err, long long limit_ticks) {
    if (threadIdx.x != 0) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;   // an earlier wait failed: drain
    const long long t0 = wall_clock64();
    // signed distance: the counters only grow and wrap after 2^32 messages
    while (static_cast<int32_t>(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > limit_ticks) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
__global__ __launch_bounds__(64) void lb_post(uint32_t* counter, uint32_t value) {
    // everything in front of this launch in its queue is complete (in-order queue, end-of-kernel release); publish for the other process
    if (threadIdx.x == 0) __hip_atomic_store(counter, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void lb_copy(char* __restrict__ dst, const char* __restrict__ src, size_t bytes) {
    const size_t n16 = bytes / 16;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
}
// out[i] = sum over ranks, in rank order, of contributions [nranks][REDUCE_MAX]
__global__ __launch_bounds__(256) void lb_reduce(double* out, const double* contrib, int nranks, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = contrib[i];
    for (int r = 1; r < nranks; ++r) s += contrib[static_cast<size_t>(r) * REDUCE_MAX + i];
    out[i] = s;
}

#define LB_HIP(x)                                                                                                   \
    do {                                                                                                            \
        hipError_t e_ = (x);                                                                                        \
        if (e_ != hipSuccess) {                                                                                     \
            std::fprintf(stderr, "[loopback_rccl] %s -> %s\n", #x, hipGetErrorString(e_));                         \
            return ncclUnhandledCudaError;                                                                          \
        }                                                                                                           \
    } while (0)

int launch_copy(char* dst, const char* src, size_t bytes, hipStream_t st) {
    if (!bytes) return ncclSuccess;
    const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (!aligned) {
        LB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return ncclSuccess;
    }
    const size_t n16 = bytes / 16;
    const int wg = static_cast<int>(std::min<size_t>(256, (n16 + 255) / 256 + 1));
    hipLaunchKernelGGL(lb_copy, dim3(wg), dim3(256), 0, st, dst, src, bytes);
    LB_HIP(hipGetLastError());
    return ncclSuccess;
}

// one chunk of a message: push into the destination's mailbox / pull out of mine
int push_chunk(Comm* c, int dst, const char* src, size_t bytes, hipStream_t st) {
    const uint32_t seq = c->sent[dst]++;
    char* b = c->box[dst];
    uint32_t* consumed = reinterpret_cast<uint32_t*>(b + c->lay.consumed(c->rank));
    uint32_t* posted = reinterpret_cast<uint32_t*>(b + c->lay.posted(c->rank));
    if (seq >= RING) {   // the slot's previous tenant (chunk seq - RING) must have been pulled
        hipLaunchKernelGGL(lb_wait, dim3(1), dim3(64), 0, st, consumed, seq - RING + 1, c->err, c->limit_ticks);
        LB_HIP(hipGetLastError());
    }
    if (int rc = launch_copy(b + c->lay.slot(c->rank, seq), src, bytes, st)) return rc;
    hipLaunchKernelGGL(lb_post, dim3(1), dim3(64), 0, st, posted, seq + 1);
    LB_HIP(hipGetLastError());
    return ncclSuccess;
}
int pull_chunk(Comm* c, int src, char* dst, size_t bytes, hipStream_t st) {
    const uint32_t seq = c->rcvd[src]++;
    uint32_t* consumed = reinterpret_cast<uint32_t*>(c->mine + c->lay.consumed(src));
    uint32_t* posted = reinterpret_cast<uint32_t*>(c->mine + c->lay.posted(src));
    hipLaunchKernelGGL(lb_wait, dim3(1), dim3(64), 0, st, posted, seq + 1, c->err, c->limit_ticks);
    LB_HIP(hipGetLastError());
    if (int rc = launch_copy(dst, c->mine + c->lay.slot(src, seq), bytes, st)) return rc;
    hipLaunchKernelGGL(lb_post, dim3(1), dim3(64), 0, st, consumed, seq + 1);
    LB_HIP(hipGetLastError());
    return ncclSuccess;
}

// A group: round k moves chunk k of every message -- all pushes of the round first, then all pulls, so that no rank's queue waits for a
// peer's push behind one of its own pulls (what "sends and receives of a group proceed together" needs with in-order queues), and never more
// than one chunk per pair is outstanding per round (the ring of RING slots cannot fill up inside a group).
int run_ops(std::vector<Op>& ops) {
    size_t rounds = 0;
    for (const Op& o : ops) {
        if (o.comm->err && *o.comm->err) return ncclSystemError;
        const size_t sb = o.comm->lay.slot_bytes;
        rounds = std::max(rounds, o.bytes ? (o.bytes + sb - 1) / sb : 1);
    }
    for (size_t k = 0; k < rounds; ++k) {
        for (int phase = 0; phase < 2; ++phase) {
            for (const Op& o : ops) {
                if (o.send != (phase == 0)) continue;
                const size_t sb = o.comm->lay.slot_bytes;
                const size_t nchunks = o.bytes ? (o.bytes + sb - 1) / sb : 1;
                if (k >= nchunks) continue;
                const size_t off = k * sb;
                const size_t len = std::min(sb, o.bytes - off);
                const int rc = o.send ? push_chunk(o.comm, o.peer, o.buf + off, len, o.stream) : pull_chunk(o.comm, o.peer, o.buf + off, len, o.stream);
                if (rc) return rc;
            }
        }
    }
    return ncclSuccess;
}

int submit(const Op& op) {
    if (g_depth > 0) {
        g_ops.push_back(op);
        return ncclSuccess;
    }
    std::vector<Op> one{op};
    return run_ops(one);
}

bool wait_for(const std::function<bool()>& cond, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!cond()) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return true;
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loopback: HIP call failed";
        case ncclSystemError: return "loopback: rendezvous / device-side wait timed out";
        case ncclInvalidArgument: return "loopback: invalid argument";
        case ncclInvalidUsage: return "loopback: invalid usage";
        default: return "loopback: internal error";
    }
}

int ncclGetUniqueId(UniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    id->magic = MAGIC;
    id->pid = static_cast<uint32_t>(getpid());
    id->nonce = static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count()) ^ (static_cast<uint64_t>(getpid()) << 40);
    std::snprintf(id->name, sizeof(id->name), "/tm_loopback_%08x_%016llx", id->pid, static_cast<unsigned long long>(id->nonce));
    return ncclSuccess;   // the segment itself is made by rank 0 in ncclCommInitRank (an id that is never used leaves nothing behind)
}

int ncclCommInitRank(Comm** out, int nranks, UniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.magic != MAGIC) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->name = id.name;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(id.name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) {
            if (fd >= 0) close(fd);
            shm_unlink(id.name);
            delete c;
            return ncclSystemError;
        }
    } else if (!wait_for([&]() { fd = shm_open(id.name, O_RDWR, 0600); return fd >= 0; }, 120.0)) {
        delete c;
        return ncclSystemError;
    }
    if (rank != 0) {   // the creator sizes the segment right after making it: wait until it has
        struct stat sb;
        if (!wait_for([&]() { return fstat(fd, &sb) == 0 && static_cast<size_t>(sb.st_size) >= sizeof(Shared); }, 60.0)) {
            close(fd);
            delete c;
            return ncclSystemError;
        }
    }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->sh = static_cast<Shared*>(p);
    c->sh_bytes = sizeof(Shared);
    if (rank == 0) {   // fresh shm is zero-filled
        const char* kb = std::getenv("TM_LOOPBACK_SLOT_KB");
        c->sh->slot_bytes = static_cast<uint64_t>(kb ? std::max(1, std::atoi(kb)) : 512) * 1024;
        c->sh->nranks = nranks;
        c->sh->created.store(1, std::memory_order_release);
    }
    if (!wait_for([&]() { return c->sh->created.load(std::memory_order_acquire) == 1; }, 60.0)) return ncclSystemError;
    c->lay.slot_bytes = c->sh->slot_bytes;
    const char* lim = std::getenv("TM_LOOPBACK_WAIT_S");
    c->limit_ticks = static_cast<long long>((lim ? std::atof(lim) : 30.0) * 1e8);
    LB_HIP(hipMalloc(reinterpret_cast<void**>(&c->mine), c->lay.total()));
    LB_HIP(hipMemset(c->mine, 0, 64 * 2 * MAX_RANKS));
    LB_HIP(hipMalloc(reinterpret_cast<void**>(&c->red), sizeof(double) * REDUCE_MAX * nranks));
    LB_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->err), sizeof(uint32_t), hipHostMallocDefault));
    *c->err = 0;
    LB_HIP(hipDeviceSynchronize());
    RankSlot& me = c->sh->r[rank];
    me.pid = static_cast<int>(getpid());
    if (nranks > 1) LB_HIP(hipIpcGetMemHandle(&me.mailbox, c->mine));
    me.ready.store(1, std::memory_order_release);
    c->box[rank] = c->mine;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        RankSlot& o = c->sh->r[r];
        if (!wait_for([&]() { return o.ready.load(std::memory_order_acquire) == 1; }, 120.0)) return ncclSystemError;
        if (o.pid == me.pid) return ncclInvalidUsage;   // two ranks of one communicator in one process: hipIpc cannot map a process's own allocation
        void* q = nullptr;
        LB_HIP(hipIpcOpenMemHandle(&q, o.mailbox, hipIpcMemLazyEnablePeerAccess));
        c->box[r] = static_cast<char*>(q);
    }
    *out = c;
    return ncclSuccess;
}

int ncclCommDestroy(Comm* c) {
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();   // everything this rank enqueued has run: its pushes are in the peers' mailboxes, its pulls are done
    if (c->sh) {
        c->sh->r[c->rank].departed.store(1, std::memory_order_release);
        // nobody frees a mailbox a peer may still push into or poll
        wait_for([&]() {
            for (int r = 0; r < c->nranks; ++r)
                if (c->sh->r[r].departed.load(std::memory_order_acquire) != 1) return false;
            return true;
        }, 30.0);
    }
    for (int r = 0; r < c->nranks; ++r)
        if (r != c->rank && c->box[r]) (void)hipIpcCloseMemHandle(c->box[r]);
    if (c->mine) (void)hipFree(c->mine);
    if (c->red) (void)hipFree(c->red);
    if (c->err) (void)hipHostFree(c->err);
    if (c->sh) {
        if (c->rank == 0) shm_unlink(c->name.c_str());
        munmap(c->sh, c->sh_bytes);
    }
    delete c;
    return ncclSuccess;
}

int ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

int ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, Comm* c, hipStream_t stream) {
    if (!c || dtype != ncclFloat64 || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    return submit(Op{true, const_cast<char*>(static_cast<const char*>(buf)), count * sizeof(double), peer, c, stream});
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, Comm* c, hipStream_t stream) {
    if (!c || dtype != ncclFloat64 || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    return submit(Op{false, static_cast<char*>(buf), count * sizeof(double), peer, c, stream});
}

// in place or not, sum of doubles: all-gather through the mailboxes, then one reduction kernel in rank order
int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, Comm* c, hipStream_t stream) {
    if (!c || dtype != ncclFloat64 || op != ncclSum || count > REDUCE_MAX) return ncclInvalidArgument;
    if (g_depth > 0) return ncclInvalidUsage;   // not needed by tm_rccl.cpp
    if (c->err && *c->err) return ncclSystemError;
    const size_t bytes = count * sizeof(double);
    if (c->nranks == 1) {
        if (sendbuf != recvbuf) LB_HIP(hipMemcpyAsync(recvbuf, sendbuf, bytes, hipMemcpyDeviceToDevice, stream));
        return ncclSuccess;
    }
    std::vector<Op> ops;
    char* red = reinterpret_cast<char*>(c->red);
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        ops.push_back(Op{true, const_cast<char*>(static_cast<const char*>(sendbuf)), bytes, r, c, stream});
        ops.push_back(Op{false, red + sizeof(double) * REDUCE_MAX * r, bytes, r, c, stream});
    }
    if (int rc = launch_copy(red + sizeof(double) * REDUCE_MAX * c->rank, static_cast<const char*>(sendbuf), bytes, stream)) return rc;
    if (int rc = run_ops(ops)) return rc;
    hipLaunchKernelGGL(lb_reduce, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, stream, static_cast<double*>(recvbuf), c->red, c->nranks, count);
    LB_HIP(hipGetLastError());
    return ncclSuccess;
}

// test support: has a device-side wait of this communicator run into its limit?
int tm_loopback_error(Comm* c) { return (c && c->err) ? static_cast<int>(*c->err) : 0; }

}  // extern "C"
