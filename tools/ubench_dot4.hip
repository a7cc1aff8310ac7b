// Issue rate of v_dot4_i32_i8 against v_fma_f64 and v_add_u32 on one SIMD (gfx950), 1 / 2 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_dot4.hip -o /tmp/ubench_dot4 && /tmp/ubench_dot4
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 100000
template <int MIX>
__global__ void k(int* out, int a, int b, double fa, double fb) {
    int x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    double d0 = fa + threadIdx.x, d1 = fa * 2, d2 = fa * 3, d3 = fa * 4, d4 = fa * 5, d5 = fa * 6, d6 = fa * 7, d7 = fa * 8;
#pragma unroll 4
    for (int i = 0; i < N_IT; ++i) {
        if (MIX == 0) {   // 8 independent dot4 chains
            x0 = __builtin_amdgcn_sdot4(x0, b, x0, false); x1 = __builtin_amdgcn_sdot4(x1, b, x1, false);
            x2 = __builtin_amdgcn_sdot4(x2, b, x2, false); x3 = __builtin_amdgcn_sdot4(x3, b, x3, false);
            x4 = __builtin_amdgcn_sdot4(x4, b, x4, false); x5 = __builtin_amdgcn_sdot4(x5, b, x5, false);
            x6 = __builtin_amdgcn_sdot4(x6, b, x6, false); x7 = __builtin_amdgcn_sdot4(x7, b, x7, false);
        } else if (MIX == 1) {   // 8 integer adds
            x0 += b ^ x1; x1 += b ^ x2; x2 += b ^ x3; x3 += b ^ x4; x4 += b ^ x5; x5 += b ^ x6; x6 += b ^ x7; x7 += b ^ x0;
        } else {   // 8 fp64 FMAs
            d0 = __builtin_fma(d0, fb, fa); d1 = __builtin_fma(d1, fb, fa); d2 = __builtin_fma(d2, fb, fa); d3 = __builtin_fma(d3, fb, fa);
            d4 = __builtin_fma(d4, fb, fa); d5 = __builtin_fma(d5, fb, fa); d6 = __builtin_fma(d6, fb, fa); d7 = __builtin_fma(d7, fb, fa);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (int)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
int main() {
    int* out;
    hipMalloc(&out, 4 * 1024 * 64);
    const char* names[3] = {"v_dot4_i32_i8", "v_xor + v_add_u32 (2 instr)", "v_fma_f64"};
    for (int mix = 0; mix < 3; ++mix)
        for (int waves : {4, 8, 16}) {   // 1 / 2 / 4 per SIMD
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                if (mix == 0) k<0><<<1, waves * 64>>>(out, 3, 0x01020304, 1.0000001, 0.9999999);
                else if (mix == 1) k<1><<<1, waves * 64>>>(out, 3, 0x01020304, 1.0000001, 0.9999999);
                else k<2><<<1, waves * 64>>>(out, 3, 0x01020304, 1.0000001, 0.9999999);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double per_simd = waves / 4.0;
            const double n = (mix == 1 ? 16.0 : 8.0) * N_IT * per_simd;   // instructions one SIMD issued
            printf("%-28s %d wave(s) per SIMD: %.3f ms -> %.2f ns per instruction of the SIMD (2.4 GHz: %.1f cycles)\n",
                   names[mix], (int)per_simd, best, best * 1e6 / n, best * 1e6 / n * 2.4);
        }
    return 0;
}
