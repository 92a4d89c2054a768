// STREAM-style ceilings on one MI355X: variants of a 16 B/lane copy (grid shape, unroll, contiguous vs strided trips, cache policy).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream tools/ubench/stream.hip ; run: /tmp/stream [MiB per array = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UN, bool NTL, bool NTS, bool CONTIG>
__global__ __launch_bounds__(256) void k_copy(d2v* __restrict__ a, const d2v* __restrict__ b, long n) {
    const long stride = CONTIG ? 256 : (long)gridDim.x * 256;
    const long step = CONTIG ? (long)gridDim.x * 256 * UN : stride * UN;
    for (long i0 = CONTIG ? (long)blockIdx.x * 256 * UN + threadIdx.x : (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += step) {
        d2v x[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const long i = i0 + q * stride < n ? i0 + q * stride : n - 1;
            x[q] = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const long i = i0 + q * stride;
            if (i < n) { if (NTS) __builtin_nontemporal_store(x[q], a + i); else a[i] = x[q]; }
        }
    }
}
static bool g_pingpong = false;   // alternate direction every launch (what a relaxation sweep does with its two fields)
template <int UN, bool NTL, bool NTS, bool CONTIG>
void run(const char* name, d2v* a, d2v* b, long n, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_copy<UN, NTL, NTS, CONTIG>), dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int k = 0; k < it; ++k) {
        if (g_pingpong && (k & 1)) hipLaunchKernelGGL((k_copy<UN, NTL, NTS, CONTIG>), dim3(grid), dim3(256), 0, 0, b, a, n);
        else hipLaunchKernelGGL((k_copy<UN, NTL, NTS, CONTIG>), dim3(grid), dim3(256), 0, 0, a, b, n);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %6d: %7.1f us  %6.0f GB/s\n", name, grid, 1e3 * ms / it, 2.0 * 16 * n * it / (1e-3 * ms) / 1e9);
}
int main(int argc, char** argv) {
    const long mib = argc > 1 ? atol(argv[1]) : 256, n = mib * 1024 * 1024 / 16;
    d2v *a, *b; CK(hipMalloc(&a, 16 * n)); CK(hipMalloc(&b, 16 * n)); CK(hipMemset(a, 0, 16 * n)); CK(hipMemset(b, 0, 16 * n));
    g_pingpong = argc > 2 && atoi(argv[2]) != 0;
    printf("%ld MiB per array, %s\n", mib, g_pingpong ? "ping-pong (a->b, b->a, ...)" : "same direction every launch");
    for (int grid : {2048, 8192, 65536}) {
        run<4, true, true, false>("strided un4 ntl nts", a, b, n, grid);
        run<4, false, true, false>("strided un4 ld nts", a, b, n, grid);
        run<8, true, true, false>("strided un8 ntl nts", a, b, n, grid);
        run<4, true, true, true>("contig un4 ntl nts", a, b, n, grid);
        run<8, true, true, true>("contig un8 ntl nts", a, b, n, grid);
        run<8, false, true, true>("contig un8 ld nts", a, b, n, grid);
        run<8, false, false, true>("contig un8 ld st", a, b, n, grid);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemcpyAsync(a, b, 16 * n, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 20; ++k) CK(hipMemcpyAsync(a, b, 16 * n, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemcpyAsync D2D: %7.1f us  %6.0f GB/s\n", 1e3 * ms / 20, 2.0 * 16 * n * 20 / (1e-3 * ms) / 1e9);
    return 0;
}
