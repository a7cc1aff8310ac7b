// (round 6 diagnosis) What is an s_memtime tick, and what do the SIMDs clock at?  One lone wave runs dependent chains of
// known length (v_fma_f64, v_add_f32, s_nop 15 = 16 issue cycles, s_sleep 8 = 512 cycles) and brackets each with BOTH
// s_memtime (what every "cycle" figure of DESIGN.md section 4.1 is counted in) and s_memrealtime (the constant 100 MHz
// counter).  If s_memtime is the shader clock, ticks per instruction are integers whatever the clock; if it is a
// fixed-rate counter, ticks per instruction move with DVFS and the nanoseconds per instruction give the real clock.
// Built as a small shared library (tools/clock_probe.py calls it alone and BESIDE trk3_kernel, which leaves 96 CUs idle):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench_clock.hip -o tools/bin/libclk.so
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAIN 64
#define REPS 256

__device__ __forceinline__ unsigned long long memtime() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }

// out[4 * k + 0 .. 3] = ticks, realtime ticks (10 ns), instructions, 0  for chain k
__global__ __launch_bounds__(64) void clk_kernel(unsigned long long* out, double a, double b, int spin) {
    double v = a + 1e-3 * threadIdx.x;
    float f = (float)b + threadIdx.x;
    unsigned long long t0, t1, r0, r1;
    // k = 0: dependent v_fma_f64
    t0 = memtime(); r0 = realtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < CHAIN; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));
    }
    asm volatile("s_nop 0" : "+v"(v));
    t1 = memtime(); r1 = realtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)REPS * CHAIN; }
    // k = 1: dependent v_add_f32
    t0 = memtime(); r0 = realtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < CHAIN; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(1.0f));
    }
    asm volatile("s_nop 0" : "+v"(f));
    t1 = memtime(); r1 = realtime();
    if (threadIdx.x == 0) { out[4] = t1 - t0; out[5] = r1 - r0; out[6] = (unsigned long long)REPS * CHAIN; }
    // k = 2: s_nop 15 (16 issue cycles each)
    t0 = memtime(); r0 = realtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < CHAIN; ++i) asm volatile("s_nop 15");
    }
    t1 = memtime(); r1 = realtime();
    if (threadIdx.x == 0) { out[8] = t1 - t0; out[9] = r1 - r0; out[10] = (unsigned long long)REPS * CHAIN * 16; }
    // k = 3: s_sleep 8 (8 x 64 cycles each)
    t0 = memtime(); r0 = realtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("s_sleep 8");
    }
    t1 = memtime(); r1 = realtime();
    if (threadIdx.x == 0) { out[12] = t1 - t0; out[13] = r1 - r0; out[14] = (unsigned long long)REPS * 8 * 512; }
    // k = 4: a long stretch of the fp64 chain (spin x REPS x CHAIN instructions): the clock over ~100 us .. ms
    t0 = memtime(); r0 = realtime();
    for (int s = 0; s < spin; ++s)
        for (int r = 0; r < REPS; ++r) {
#pragma unroll
            for (int i = 0; i < CHAIN; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));
        }
    asm volatile("s_nop 0" : "+v"(v));
    t1 = memtime(); r1 = realtime();
    if (threadIdx.x == 0) { out[16] = t1 - t0; out[17] = r1 - r0; out[18] = (unsigned long long)spin * REPS * CHAIN; }
    if (v == 12345.678 && f == 3.0f) out[31] = 1;   // (keeps the chains alive)
}

// every CU busy with fp64 (n_wg workgroups of 256 threads) for about `spin` x 0.1 ms: what load does to the clock
__global__ __launch_bounds__(256) void load_kernel(double* sink, double a, double b, int spin) {
    double v = a + 1e-3 * threadIdx.x, w = b;
    for (int s = 0; s < spin; ++s)
        for (int r = 0; r < REPS * 8; ++r) {
            v = __builtin_fma(v, b, a);
            w = __builtin_fma(w, a, b);
        }
    if (v + w == 12345.678) sink[0] = v;
}

static hipStream_t g_st = nullptr, g_st2 = nullptr;
static unsigned long long* g_out = nullptr;   // pinned host memory: the kernel writes it directly
static double* g_sink = nullptr;

extern "C" int clk_probe(unsigned long long* res /* [20] */, int spin, double* event_ms) {
    if (!g_st) {
        if (hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipHostMalloc((void**)&g_out, 32 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return -2;
    }
    for (int i = 0; i < 32; ++i) g_out[i] = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, g_st);
    clk_kernel<<<1, 64, 0, g_st>>>(g_out, 1e-9, 0.999999, spin);
    hipEventRecord(e1, g_st);
    if (hipStreamSynchronize(g_st) != hipSuccess) return -3;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (event_ms) *event_ms = ms;
    for (int i = 0; i < 20; ++i) res[i] = g_out[i];
    return 0;
}

// start `n_wg` workgroups of fp64 load on a second stream (returns at once); clk_load_wait() joins it
extern "C" int clk_load_start(int n_wg, int spin) {
    if (!g_st2) {
        if (hipStreamCreateWithFlags(&g_st2, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipMalloc((void**)&g_sink, 64) != hipSuccess) return -2;
    }
    load_kernel<<<n_wg, 256, 0, g_st2>>>(g_sink, 1e-9, 0.999999, spin);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int clk_load_wait() { return hipStreamSynchronize(g_st2) == hipSuccess ? 0 : -1; }
