// Micro-benchmark: issue rate of the VALU instructions the Winslow row uses, on gfx950.
// Each wave runs ITER iterations of 8 independent copies of one instruction; 4 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

#define KERNEL(NAME, ASM8) KERNELT(NAME, double, ASM8)
#define KERNEL32(NAME, ASM8) KERNELT(NAME, float, ASM8)
#define KERNELT(NAME, TY, ASM8)                                                                    \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                        \
        TY a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        TY b = 1.0000001, c = 0.5;                                                             \
        for (int i = 0; i < ITER; ++i) { asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456) out[0] = a0;                         \
    }

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define FMA(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n\t"
#define ADD(k) "v_add_f64 %" #k ", %" #k ", %8\n\t"
#define MUL(k) "v_mul_f64 %" #k ", %" #k ", %8\n\t"
#define RCP(k) "v_rcp_f64 %" #k ", %" #k "\n\t"
#define DSC(k) "v_div_scale_f64 %" #k ", vcc, %" #k ", %8, %" #k "\n\t"
#define DFM(k) "v_div_fmas_f64 %" #k ", %" #k ", %8, %9\n\t"
#define DFX(k) "v_div_fixup_f64 %" #k ", %" #k ", %8, %9\n\t"
#define MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define DPP(k) "v_mov_b32_dpp %" #k ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CND(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define CMP(k) "v_cmp_neq_f64 vcc, %" #k ", %8\n\t"

KERNEL(k_fma, REP8(FMA))
KERNEL(k_add, REP8(ADD))
KERNEL(k_mul, REP8(MUL))
KERNEL(k_rcp, REP8(RCP))
KERNEL(k_dsc, REP8(DSC))
KERNEL(k_dfm, REP8(DFM))
KERNEL(k_dfx, REP8(DFX))
KERNEL32(k_mov, REP8(MOV))
KERNEL32(k_dpp, REP8(DPP))
KERNEL32(k_cnd, REP8(CND))
KERNEL32(k_fma32, REP8(FMA32))
KERNEL(k_pkfma, REP8(PKFMA))
KERNEL(k_cmp, REP8(CMP))

int main() {
    double* d;
    CHECK(hipMalloc(&d, 1024));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;   // Hz
    printf("device %s, %d CUs, %.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
    struct T { const char* name; void (*fn)(double*, double); };
    T tests[] = {{"v_fma_f64", k_fma}, {"v_add_f64", k_add}, {"v_mul_f64", k_mul}, {"v_rcp_f64", k_rcp}, {"v_div_scale_f64", k_dsc}, {"v_div_fmas_f64", k_dfm},
                 {"v_div_fixup_f64", k_dfx}, {"v_mov_b32", k_mov}, {"v_mov_b32_dpp", k_dpp}, {"v_cndmask_b32", k_cnd}, {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma},
                 {"v_cmp_neq_f64", k_cmp}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 4}) {   // waves per SIMD
        const int grid = cus * wps;   // 256 threads = 4 waves = one per SIMD
        for (auto& t : tests) {
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, d, 1.0);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, d, 1.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double inst_per_simd = double(ITER) * 8 * wps;
            printf("waves/SIMD %d  %-18s %8.1f us  -> %.2f cycles per wave-instruction per SIMD\n", wps, t.name, ms * 1e3, ms * 1e-3 * clk / inst_per_simd);
        }
    }
    return 0;
}
