// Microbenchmark: per-wave-instruction issue cost (cycles) of fp64 VALU ops on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 256
template <int OP, int CHAINS>
__global__ void k(double* out, long long* cyc, double a, double b) {
    double v[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = a + i + threadIdx.x;
    int iv[CHAINS];
    for (int i = 0; i < CHAINS; ++i) iv[i] = threadIdx.x + i;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) v[i] = __builtin_fma(v[i], a, b);
            if (OP == 1) v[i] = v[i] * a;
            if (OP == 2) v[i] = v[i] + b;
            if (OP == 3) { iv[i] = (int)v[i]; v[i] = (double)(iv[i] + 1); }      // cvt both ways
            if (OP == 4) { iv[i] = iv[i] > it ? iv[i] + 3 : iv[i] ^ 5; }          // cmp+cndmask-ish int
            if (OP == 5) v[i] = ceil(v[i]) + b;
            if (OP == 6) { float f = (float)iv[i]; f = __builtin_fmaf(f, 1.5f, 2.0f); iv[i] = (int)f; }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; int si = 0;
    for (int i = 0; i < CHAINS; ++i) { s += v[i]; si += iv[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + si;
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&cyc[0], (unsigned long long)t0); atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1); }
}
template <int OP, int CHAINS>
void run(const char* name, int threads, int ops_per) {
    double* d; long long* c; (void)hipMalloc(&d, 8 * 4096); (void)hipMalloc(&c, 8 * 16);
    k<OP, CHAINS><<<1, threads>>>(d, c, 1.0000001, 1e-9);
    long long init[2] = {0x7fffffffffffffffll, 0}; (void)hipMemcpy(c, init, 16, hipMemcpyHostToDevice);
    k<OP, CHAINS><<<1, threads>>>(d, c, 1.0000001, 1e-9);
    long long hh[2]; (void)hipMemcpy(hh, c, 16, hipMemcpyDeviceToHost); long long h = hh[1] - hh[0];
    double per = (double)h / (N_IT * CHAINS * ops_per);
    int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    printf("%-28s threads %4d chains %2d : %.2f cycles per op per wave (x%d waves/SIMD => %.2f cycles/op/SIMD-slot)\n",
           name, threads, CHAINS, per, waves_per_simd, per / waves_per_simd);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    for (int th : {64, 256, 512, 1024}) {
        if (th == 64) { run<0, 8>("fma_f64", 64, 1); run<0, 1>("fma_f64 dependent", 64, 1); run<1, 8>("mul_f64", 64, 1); run<2, 8>("add_f64", 64, 1);
                        run<2, 1>("add_f64 dependent", 64, 1); run<3, 8>("cvt i32<->f64 (2 ops)", 64, 2); run<4, 8>("int cmp+sel (~3 ops)", 64, 3); run<5, 8>("ceil+add f64 (2 ops)", 64, 2);
                        run<6, 8>("f32 cvt+fma+cvt (3 ops)", 64, 3);}
        if (th == 256) { run<0, 8>("fma_f64", 256, 1); run<1, 8>("mul_f64", 256, 1); run<2, 8>("add_f64", 256, 1); }
        if (th == 512) { run<0, 8>("fma_f64", 512, 1); run<2, 8>("add_f64", 512, 1); run<0, 2>("fma_f64 2 chains", 512, 1); run<4, 8>("int cmp+sel (~3 ops)", 512, 3); run<3, 8>("cvt i32<->f64 (2 ops)", 512, 2);}
        if (th == 1024) { run<0, 8>("fma_f64", 1024, 1); run<1, 8>("mul_f64", 1024, 1); run<2, 8>("add_f64", 1024, 1); run<3, 8>("cvt i32<->f64 (2 ops)", 1024, 2); run<4, 8>("int cmp+sel (~3 ops)", 1024, 3);}
    }
    return 0;
}
