// What does this part sustain for the STREAM MIXES of the Krylov kernels?  NR read arrays + NW written arrays of 256 MiB each, 16 B per lane
// and array, no re-use, streaming loads and stores -- the traffic of k_apply_vk<VK_S2> (4 read : 1 written) and k_apply_vk<VK_R> (7 : 4)
// at 4096^2 without any of their arithmetic, halo rows or registers.  A ceiling for "bytes / time" of those kernels that a copy (1 : 1) does
// not give: the more arrays a pass walks at once, the more DRAM pages it keeps open per channel.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_mix tools/ubench/stream_mix.hip ; run: /tmp/stream_mix [MiB per array = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Arrays { d2v* p[12]; };

// a trip = UN consecutive 256-lane segments of every array: all NR * UN loads are issued before the first store
template <int NR, int NW, int UN>
__global__ __launch_bounds__(256) void k_mix(Arrays A, long n) {
    const long step = (long)gridDim.x * 256 * UN;
    for (long i0 = (long)blockIdx.x * 256 * UN + threadIdx.x; i0 < n; i0 += step) {
        d2v x[NR][UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const long i = i0 + q * 256 < n ? i0 + q * 256 : n - 1;
#pragma unroll
            for (int r = 0; r < NR; ++r) x[r][q] = __builtin_nontemporal_load(A.p[r] + i);
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const long i = i0 + q * 256;
            d2v s = x[0][q];
#pragma unroll
            for (int r = 1; r < NR; ++r) s += x[r][q];
            if (i < n) {
#pragma unroll
                for (int w = 0; w < NW; ++w) __builtin_nontemporal_store(s * (double)(w + 1), A.p[NR + w] + i);
            }
        }
    }
}

// the same with 32 B per lane and array (two vectors of a node side by side): fewer, wider streams for the same bytes
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_mix_wide(Arrays A, long n) {
    const long step = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
        d4v x[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d4v*>(A.p[r]) + i);
        d4v s = x[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) s += x[r];
#pragma unroll
        for (int w = 0; w < NW; ++w) __builtin_nontemporal_store(s * (double)(w + 1), reinterpret_cast<d4v*>(A.p[NR + w]) + i);
    }
}
template <int NR, int NW>
void run_wide(const Arrays& A, long n32, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_mix_wide<NR, NW>), dim3(grid), dim3(256), 0, 0, A, n32);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int k = 0; k < it; ++k) hipLaunchKernelGGL((k_mix_wide<NR, NW>), dim3(grid), dim3(256), 0, 0, A, n32);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %d read : %d written at 32 B per lane, grid %6d: %8.1f us  %6.0f GB/s\n", NR, NW, grid, 1e3 * ms / it,
           (double)(NR + NW) * 32 * n32 * it / (1e-3 * ms) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int NR, int NW, int UN>
double run(const Arrays& A, long n, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_mix<NR, NW, UN>), dim3(grid), dim3(256), 0, 0, A, n);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int k = 0; k < it; ++k) hipLaunchKernelGGL((k_mix<NR, NW, UN>), dim3(grid), dim3(256), 0, 0, A, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbps = (double)(NR + NW) * 16 * n * it / (1e-3 * ms) / 1e9;
    printf("  %d read : %d written, %d segments per trip, grid %6d: %8.1f us  %6.0f GB/s\n", NR, NW, UN, grid, 1e3 * ms / it, gbps);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return gbps;
}

template <int NR, int NW>
void mix(const char* what, const Arrays& A, long n) {
    printf("%s\n", what);
    double best = 0;
    for (int grid : {1024, 2048, 8192}) {
        best = std::max(best, run<NR, NW, 1>(A, n, grid));
        best = std::max(best, run<NR, NW, 2>(A, n, grid));
        if (NR <= 4) best = std::max(best, run<NR, NW, 4>(A, n, grid));
    }
    printf("  best %.0f GB/s = %.3f of 8 TB/s\n", best, best / 8000.0);
}

int main(int argc, char** argv) {
    const long mib = argc > 1 ? atol(argv[1]) : 256, n = mib * 1024 * 1024 / 16;
    Arrays A;
    for (int k = 0; k < 11; ++k) { CK(hipMalloc(&A.p[k], 16 * n)); CK(hipMemset(A.p[k], 0, 16 * n)); }
    A.p[11] = nullptr;
    printf("%ld MiB per array\n", mib);
    mix<1, 1>("copy", A, n);
    mix<2, 1>("triad", A, n);
    mix<4, 1>("k_apply_vk<VK_S2>'s mix (r, v, frozen field, r_hat -> t)", A, n);
    mix<7, 4>("k_apply_vk<VK_R>'s mix (r, v, t, p, u, frozen field, r_hat -> r', p', v', u)", A, n);
    // Do the arrays' relative positions matter?  (Same index in every array = same low address bits when the arrays lie a multiple of
    // 256 MiB apart: the channel / bank a workgroup's 11 requests of one trip map to.)  One slab, array k at k * (256 MiB + pad).
    {
        const long pads[] = {0, 256, 4096, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 3 * 65536 + 3 * 4096 + 768};
        char* slab; CK(hipMalloc(&slab, 11 * (16 * n + (2 << 20))));
        CK(hipMemset(slab, 0, 11 * (16 * n + (2 << 20))));
        for (long pad : pads) {
            Arrays B;
            for (int k = 0; k < 11; ++k) B.p[k] = reinterpret_cast<d2v*>(slab + k * (16 * n + pad));
            B.p[11] = nullptr;
            printf("one slab, arrays %ld MiB + %ld B apart\n", mib, pad);
            run<7, 4, 2>(B, n, 8192);
            run<7, 4, 2>(B, n, 1024);
            run<4, 1, 1>(B, n, 2048);
        }
        CK(hipFree(slab));
    }
    printf("the same mixes as fewer, wider streams (256 MiB per array, 32 B per lane)\n");
    for (int grid : {1024, 2048, 8192}) run_wide<2, 1>(A, n / 2, grid);   // ~ VK_S2 with (r, v) side by side: 4 x 16 B read, 2 x 16 B written
    for (int grid : {1024, 2048, 8192}) run_wide<4, 2>(A, n / 2, grid);   // ~ VK_R: 8 x 16 B read, 4 x 16 B written in 6 streams
    for (int grid : {1024, 2048, 8192}) run_wide<3, 2>(A, n / 2, grid);
    return 0;
}
