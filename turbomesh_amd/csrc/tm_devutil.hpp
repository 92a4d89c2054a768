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

// Per-workgroup partials: every lane holds acc[]; sums waves in fixed order -> deterministic.  NP = how many of the MAX_PARTIALS
// columns the caller's reduction uses: only those are summed (a wave sum is six dependent cross-lane steps), the rest of the
// row is written as zeros.
template <int NT, int NP = MAX_PARTIALS>
__device__ __forceinline__ void block_partials(double (&acc)[MAX_PARTIALS], double* dst) {
    constexpr int NW = NT / 64;
    __shared__ double sh[NW][NP > 0 ? NP : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) sh[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < MAX_PARTIALS) {
        double s = 0.0;
        if (static_cast<int>(threadIdx.x) < NP) {
#pragma unroll
            for (int w = 0; w < NW; ++w) s += sh[w][threadIdx.x];
        }
        dst[threadIdx.x] = s;
    }
}

// columns of a partial row each reduction flavour uses (tm_kernels.h: DotKind)
__host__ __device__ constexpr int dot_columns(int dot) {
    return dot == DOT_NONE ? 0 : (dot == DOT_AUX || dot == DOT_OUT2 || dot == DOT_DELTA) ? 2 : (dot == DOT_IN_SS) ? 6 : (dot == DOT_B2) ? 8 : 4;
}

}  // namespace tmh
