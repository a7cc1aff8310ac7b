// fp64 issue rate of a SIMD against the number of waves resident on it (gfx950): every wave runs the same mix the
// tracking kernel's map waves run (independent fp64 FMA chains, or FMA + integer mask pairs).  Trust the WALL time:
// measured 2.25 ns per wave-instruction with one wave per SIMD, 1.87 / 1.82 / 1.78 ns aggregate with 2 / 3 / 4 - a lone
// wave already gets 80 % of the SIMD's fp64 rate (4.3 cycles per wave64 instruction), so spreading the map over more
// waves of the same CU buys nothing (tried: one wave group per correlator arm, 12 map waves - no faster).  s_memtime
// differences taken inside the waves do NOT show the contention (they read 5.4 "cycles" whatever the occupancy).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 200000
template <int MIX>
__global__ void issue_kernel(double* out, long long* cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N_IT; ++i) {
        if (MIX == 0) {   // 8 independent chains
            x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
            x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
        } else {          // 4 chains of FMA + high-dword mask (the accumulation loop's pattern)
            x0 = __builtin_fma(x0, b, a); x1 = __hiloint2double(__double2hiint(x1) & (i | 0x7FF00000), __double2loint(x1));
            x2 = __builtin_fma(x2, b, a); x3 = __hiloint2double(__double2hiint(x3) & (i | 0x7FF00000), __double2loint(x3));
            x4 = __builtin_fma(x4, b, a); x5 = __hiloint2double(__double2hiint(x5) & (i | 0x7FF00000), __double2loint(x5));
            x6 = __builtin_fma(x6, b, a); x7 = __hiloint2double(__double2hiint(x7) & (i | 0x7FF00000), __double2loint(x7));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * 64); hipMalloc(&cyc, 8 * 1024);
    for (int mix = 0; mix < 2; ++mix)
        for (int waves : {1, 4, 8, 12, 16}) {
            long long h[16];
            for (int rep = 0; rep < 2; ++rep) {
                if (mix == 0) issue_kernel<0><<<1, waves * 64>>>(out, cyc, 1.0000001, 0.9999999);
                else issue_kernel<1><<<1, waves * 64>>>(out, cyc, 1.0000001, 0.9999999);
                hipDeviceSynchronize();
            }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (mix == 0) issue_kernel<0><<<1, waves * 64>>>(out, cyc, 1.0000001, 0.9999999);
            else issue_kernel<1><<<1, waves * 64>>>(out, cyc, 1.0000001, 0.9999999);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("   wall %.3f ms -> %.2f ns per wave-instruction of one wave; ", ms, ms * 1e6 / (N_IT * 8.0));
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            const double per = (double)h[0] / (N_IT * 8);
            const int per_simd = (waves + 3) / 4;
            printf("mix %d  %2d waves in the workgroup (%d per SIMD): %5.2f cycles per wave-instruction for one wave, %5.2f for the SIMD (s_memtime ticks are 100 MHz x ?: see ratio)\n",
                   mix, waves, per_simd, per, per / per_simd);
        }
    return 0;
}
