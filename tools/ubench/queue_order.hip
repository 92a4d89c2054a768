// Cross-queue ordering on one GPU: what does a dependency between two HIP streams cost, by mechanism?
//   events   hipEventRecord + hipStreamWaitEvent                        (barrier packets)
//   spin     one-wave kernels: announce (atomic add) / wait (poll)      (what libtm_hip's sweep schedule uses)
//   value    hipStreamWriteValue32 / hipStreamWaitValue32 on SIGNAL memory (hipExtMallocWithFlags(hipMallocSignalMemory)): queue packets,
//            no wave -- and on plain device memory, where the runtime substitutes its own polling kernel
// A ping-pong of N hops between a default-priority and a high-priority stream with a small kernel behind every hop; us per hop.
// Also: the co-residency self-test (a bounded wait for a kernel enqueued LATER on the other stream) and what it reads when both
// streams share one hardware queue (run with GPU_MAX_HW_QUEUES=1).
// build: hipcc --offload-arch=gfx950 -O2 -o queue_order_bin queue_order.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
            std::exit(1);                                                                         \
        }                                                                                         \
    } while (0)

__global__ void k_work(double* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
__global__ void k_signal(unsigned* c) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded by the 100 MHz constant clock: limit_ticks / 1e8 seconds
__global__ void k_wait(const unsigned* c, unsigned target, unsigned* err, long long limit_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > limit_ticks) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int hops = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int work = argc > 2 ? std::atoi(argv[2]) : 1 << 16;
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d, GPU_MAX_HW_QUEUES = %s\n", can, std::getenv("GPU_MAX_HW_QUEUES") ? std::getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    hipStream_t a, b;
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
    double* buf;
    CHECK(hipMalloc(&buf, sizeof(double) * work * 2));
    CHECK(hipMemset(buf, 0, sizeof(double) * work * 2));
    unsigned* flags;
    CHECK(hipMalloc(&flags, 64));
    unsigned h[4];
    const int wg = (work + 255) / 256;

    // ---- the co-residency self-test: a waits (2 ms limit) for a kernel enqueued on b AFTER the waiter
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(flags, 0, 64));
        CHECK(hipDeviceSynchronize());
        const double t0 = now_us();
        k_wait<<<1, 64, 0, a>>>(flags, 1u, flags + 2, 200000);   // 2 ms
        k_signal<<<1, 64, 0, b>>>(flags);
        CHECK(hipDeviceSynchronize());
        const double t1 = now_us();
        CHECK(hipMemcpy(h, flags, 16, hipMemcpyDeviceToHost));
        std::printf("self-test %d: waiter on a, announcer enqueued later on b: counter %u error %u, %.0f us wall -> %s\n", rep, h[0], h[2], t1 - t0,
                    h[2] ? "streams SHARE a queue (or b cannot start beside a): use events" : "streams run side by side");
    }

    // the same between two streams of EQUAL priority (with GPU_MAX_HW_QUEUES=1 they share the one hardware queue of that priority)
    {
        hipStream_t c2;
        CHECK(hipStreamCreateWithFlags(&c2, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(flags, 0, 64));
            CHECK(hipDeviceSynchronize());
            const double t0 = now_us();
            k_wait<<<1, 64, 0, a>>>(flags, 1u, flags + 2, 200000);   // 2 ms
            k_signal<<<1, 64, 0, c2>>>(flags);
            CHECK(hipDeviceSynchronize());
            const double t1 = now_us();
            CHECK(hipMemcpy(h, flags, 16, hipMemcpyDeviceToHost));
            std::printf("self-test %d, equal priorities: counter %u error %u, %.0f us wall -> %s\n", rep, h[0], h[2], t1 - t0,
                        h[2] ? "streams SHARE a queue: use events" : "streams run side by side");
        }
        CHECK(hipStreamDestroy(c2));
    }

    // ---- events
    hipEvent_t ea, eb;
    CHECK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    auto run = [&](const char* name, auto&& hop) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(flags, 0, 64));
            CHECK(hipDeviceSynchronize());
            const double t0 = now_us();
            for (int i = 1; i <= hops; ++i) hop(i);
            CHECK(hipDeviceSynchronize());
            const double t1 = now_us();
            if (rep == 1) std::printf("%-34s %8.2f us per round trip (work kernel on each side + 2 hops)\n", name, (t1 - t0) / hops);
        }
    };
    // baseline: the same kernels with no cross-stream ordering at all, and on one stream
    run("no ordering, two streams", [&](int) {
        k_work<<<wg, 256, 0, a>>>(buf, work);
        k_work<<<wg, 256, 0, b>>>(buf + work, work);
    });
    run("one stream", [&](int) {
        k_work<<<wg, 256, 0, a>>>(buf, work);
        k_work<<<wg, 256, 0, a>>>(buf + work, work);
    });
    run("events", [&](int) {
        k_work<<<wg, 256, 0, a>>>(buf, work);
        CHECK(hipEventRecord(ea, a));
        CHECK(hipStreamWaitEvent(b, ea, 0));
        k_work<<<wg, 256, 0, b>>>(buf + work, work);
        CHECK(hipEventRecord(eb, b));
        CHECK(hipStreamWaitEvent(a, eb, 0));
    });
    run("spin kernels (signal / wait)", [&](int i) {
        k_work<<<wg, 256, 0, a>>>(buf, work);
        k_signal<<<1, 64, 0, a>>>(flags);
        k_wait<<<1, 64, 0, b>>>(flags, (unsigned)i, flags + 2, 100000000);
        k_work<<<wg, 256, 0, b>>>(buf + work, work);
        k_signal<<<1, 64, 0, b>>>(flags + 1);
        k_wait<<<1, 64, 0, a>>>(flags + 1, (unsigned)i, flags + 2, 100000000);
    });
    if (can) {
        unsigned* sig = nullptr;
        hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory);   // (64 B: invalid argument -- a signal is exactly 8 bytes)
        std::printf("hipExtMallocWithFlags(hipMallocSignalMemory, 8 B) -> %s\n", hipGetErrorString(e));
        unsigned *sa = nullptr, *sb = nullptr;
        if (e == hipSuccess) {
            // signal memory is one 8-byte HSA signal value per allocation: two allocations
            sa = sig;
            hipError_t e2 = hipExtMallocWithFlags(reinterpret_cast<void**>(&sb), 8, hipMallocSignalMemory);
            std::printf("second signal allocation -> %s\n", hipGetErrorString(e2));
            if (e2 == hipSuccess) {
                CHECK(hipStreamWriteValue32(a, sa, 0, 0));
                CHECK(hipStreamWriteValue32(b, sb, 0, 0));
                CHECK(hipDeviceSynchronize());
                for (int rep = 0; rep < 2; ++rep) {
                    const int base = rep * hops;
                    CHECK(hipDeviceSynchronize());
                    const double t0 = now_us();
                    for (int i = 1; i <= hops; ++i) {
                        k_work<<<wg, 256, 0, a>>>(buf, work);
                        CHECK(hipStreamWriteValue32(a, sa, base + i, 0));
                        CHECK(hipStreamWaitValue32(b, sa, base + i, hipStreamWaitValueGte, 0xFFFFFFFFu));
                        k_work<<<wg, 256, 0, b>>>(buf + work, work);
                        CHECK(hipStreamWriteValue32(b, sb, base + i, 0));
                        CHECK(hipStreamWaitValue32(a, sb, base + i, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    }
                    CHECK(hipDeviceSynchronize());
                    const double t1 = now_us();
                    if (rep == 1) std::printf("%-34s %8.2f us per round trip\n", "WriteValue32 / WaitValue32 (signal)", (t1 - t0) / hops);
                }
                // spin-kernel announce + WaitValue on signal memory?  (a kernel cannot write an HSA signal's value portably: skipped)
            }
        }
        // plain device memory: the runtime's own polling kernel
        unsigned* pm;
        CHECK(hipMalloc(&pm, 64));
        CHECK(hipMemset(pm, 0, 64));
        hipError_t ew = hipStreamWriteValue32(a, pm, 0, 0);
        std::printf("hipStreamWriteValue32 on plain hipMalloc memory -> %s\n", hipGetErrorString(ew));
        if (ew == hipSuccess) {
            CHECK(hipDeviceSynchronize());
            for (int rep = 0; rep < 2; ++rep) {
                const int base = rep * hops;
                const double t0 = now_us();
                bool ok = true;
                for (int i = 1; i <= hops && ok; ++i) {
                    k_work<<<wg, 256, 0, a>>>(buf, work);
                    ok = ok && hipStreamWriteValue32(a, pm, base + i, 0) == hipSuccess;
                    ok = ok && hipStreamWaitValue32(b, pm, base + i, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess;
                    k_work<<<wg, 256, 0, b>>>(buf + work, work);
                    ok = ok && hipStreamWriteValue32(b, pm + 8, base + i, 0) == hipSuccess;
                    ok = ok && hipStreamWaitValue32(a, pm + 8, base + i, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess;
                }
                CHECK(hipDeviceSynchronize());
                const double t1 = now_us();
                if (rep == 1) std::printf("%-34s %8.2f us per round trip%s\n", "WriteValue32 / WaitValue32 (plain)", (t1 - t0) / hops, ok ? "" : " (a call FAILED)");
            }
        }
    }
    std::printf("done\n");
    return 0;
}
