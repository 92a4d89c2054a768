// Device helpers shared by the kernel translation units (tm_kernels.hip, tm_csr.hip): wave64 shuffle reduction and the
// per-workgroup partial sums every fused reduction writes (fixed summation order -> deterministic results).
#pragma once
#include "tm_kernels.h"

namespace tmh {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

// Per-workgroup partials: every lane holds acc[]; sums waves in fixed order -> deterministic.
template <int NT>
__device__ __forceinline__ void block_partials(double (&acc)[MAX_PARTIALS], double* dst) {
    constexpr int NW = NT / 64;
    __shared__ double sh[NW][MAX_PARTIALS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MAX_PARTIALS; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) sh[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < MAX_PARTIALS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sh[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

}  // namespace tmh
