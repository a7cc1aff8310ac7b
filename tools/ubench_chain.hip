// Microbenchmarks behind the tracking kernel's per-block dependency chain:
//   A  dependent-chain latencies of one wave while the other three of its workgroup wait at a barrier
//      (fp64 fma/add, rcp/rsq/sqrt/div, libm atan, the kernel's sincos), LDS hop prices;
//   B  one round of the member-to-member exchange, 8 channels x P members, three transports:
//      tagged granules (what the kernel used in round 1), 64-bit integer atomics with arrival tags
//      at workgroup scope (same-XCD L2) and at agent scope.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_chain.hip -o /tmp/ubench_chain && /tmp/ubench_chain
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define N_IT 512

__device__ __forceinline__ double div_rn(double a, double b, double y) {
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

__device__ __forceinline__ void sincos_turns(double u, double& sn, double& cs) {
    const double q = rint(u * 4.0);
    const double f = __builtin_fma(q, -0.25, u);
    const int qi = (int)q & 3;
    const double th = f * 6.283185307179586476925287;
    const double t2 = th * th;
    double ps = -2.8114572543455206e-15;
    ps = __builtin_fma(ps, t2, 7.6471637318198164e-13);
    ps = __builtin_fma(ps, t2, -1.6059043836821613e-10);
    ps = __builtin_fma(ps, t2, 2.5052108385441720e-08);
    ps = __builtin_fma(ps, t2, -2.7557319223985893e-06);
    ps = __builtin_fma(ps, t2, 1.9841269841269841e-04);
    ps = __builtin_fma(ps, t2, -8.3333333333333332e-03);
    ps = __builtin_fma(ps, t2, 1.6666666666666666e-01);
    double pc = 4.7794773323873853e-14;
    pc = __builtin_fma(pc, t2, -1.1470745597729725e-11);
    pc = __builtin_fma(pc, t2, 2.0876756987868100e-09);
    pc = __builtin_fma(pc, t2, -2.7557319223985888e-07);
    pc = __builtin_fma(pc, t2, 2.4801587301587302e-05);
    pc = __builtin_fma(pc, t2, -1.3888888888888889e-03);
    pc = __builtin_fma(pc, t2, 4.1666666666666664e-02);
    pc = __builtin_fma(pc, t2, -0.5);
    const double s0 = __builtin_fma(-(ps * t2), th, th);
    const double c0 = __builtin_fma(pc, t2, 1.0);
    sn = (qi == 0) ? s0 : (qi == 1) ? c0 : (qi == 2) ? -s0 : -c0;
    cs = (qi == 0) ? c0 : (qi == 1) ? -s0 : (qi == 2) ? -c0 : s0;
}

// OP: 0 fma  1 add  2 rcp  3 rsq  4 v_sqrt  5 sqrt()  6 a/b  7 atan(a/b)  8 sincos_turns  9 div_rn  10 ceil
//     11 lds store->load same wave  12 lds store -> barrier -> load   13 cvt f64->i32->f64  14 readlane hop  15 atan only
template <int OP>
__global__ __launch_bounds__(256) void chain_kernel(double* out, long long* cyc, double a, double b) {
    __shared__ double s_x[64];
    const int wave = threadIdx.x >> 6;
    double v = a + 1e-3 * (threadIdx.x & 63);
    long long t0 = 0, t1 = 0;
    const bool solo = (OP != 12);
    if (solo && wave != 0) {
        __syncthreads();
        return;
    }
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N_IT; ++it) {
        if (OP == 0) v = __builtin_fma(v, a, b);
        if (OP == 1) v = v + b;
        if (OP == 2) v = __builtin_amdgcn_rcp(v) + 0.5;
        if (OP == 3) v = __builtin_amdgcn_rsq(v) + 0.5;
        if (OP == 4) v = __builtin_amdgcn_sqrt(v) + 0.5;
        if (OP == 5) v = sqrt(v) + 0.5;
        if (OP == 6) v = b / v + 1.5;
        if (OP == 7) v = atan(b / v) + 1.5;
        if (OP == 8) {
            double s, c;
            sincos_turns(v, s, c);
            v = s * 0.25 + c * 0.25 + 0.75;
        }
        if (OP == 9) v = div_rn(v, 3.14159265358979, 0.318309886183790) + 1.0;
        if (OP == 10) v = ceil(v * 1.25) * 0.75 + 0.1;
        if (OP == 11) {
            s_x[threadIdx.x & 63] = v;
            v = s_x[(threadIdx.x + 1) & 63] + b;
        }
        if (OP == 12) {
            if (wave == (it & 3)) s_x[threadIdx.x & 63] = v;
            __syncthreads();
            v = s_x[(threadIdx.x + 1) & 63] + b;
        }
        if (OP == 13) v = (double)((int)v + 1) * 0.999;
        if (OP == 14) {
            const int lo = __double2loint(v), hi = __double2hiint(v);
            v = __hiloint2double(__builtin_amdgcn_readlane(hi, 5), __builtin_amdgcn_readlane(lo, 5)) + b * threadIdx.x;
        }
        if (OP == 15) v = atan(v) + 1.5;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (solo) __syncthreads();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run_chain(const char* name, double a, double b) {
    double* d;
    long long* c;
    (void)hipMalloc(&d, 8 * 256);
    (void)hipMalloc(&c, 64);
    chain_kernel<OP><<<1, 256>>>(d, c, a, b);
    chain_kernel<OP><<<1, 256>>>(d, c, a, b);
    long long h = 0;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("A  %-44s %8.1f cycles per dependent step\n", name, (double)h / N_IT);
    (void)hipFree(d);
    (void)hipFree(c);
}

// ---- B: exchange rounds ------------------------------------------------------------------------
#define MAXP 16
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

// MODE 0: granules (12 per member, one plain/agent 8-byte store each; every member polls all)
// MODE 1: integer atomics, workgroup scope (L2 of the XCD), sc1 polls
// MODE 2: integer atomics, agent scope
// MODE 3: as 1, the six words spread over two 64-byte lines (I/Q of P | E and L)
// MODE 4: as 1, polled with returning atomic ORs of zero at workgroup scope (always served by the XCD's L2)
// MODE 5: as 1, polled with nt loads
template <int MODE>
__global__ __launch_bounds__(256) void xch_kernel(unsigned long long* xch, long long* cyc, int P, int rounds, int work,
                                                  double* sink, int* bad, long long stride_words) {
    const int bq = blockIdx.x >> 3, br = blockIdx.x & 7;
    const int ch = br + 8 * (bq / P);
    const int member = bq % P;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long* base = xch + (size_t)ch * stride_words;
    __shared__ double s_v[8];
    double acc = 1.0 + member;
    long long t_total = 0;
    unsigned long long prev[2] = {0, 0};
    for (int it = 0; it < rounds; ++it) {
        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) break;   // a member timed out
        // stand-in for the map phase: `work` dependent fmas on every lane
        for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, 1e-9);
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        double tot = 0.0;
        if (MODE == 0) {
            if (wave == 0 && lane < 6) {
                const double val = (double)(member + 1) * (lane + 1) + it;
                const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
                unsigned long long* gp = base + ((it & 1) * MAXP + member) * 12 + 2 * lane;
                __hip_atomic_store(gp, tag | (unsigned)__double2loint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(gp + 1, tag | (unsigned)__double2hiint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave == 0) {
                const unsigned epoch = (unsigned)(it + 1);
                const int row = lane >> 4, c = lane & 15;
                const bool mine = c < P;
                const unsigned long long* gp = base + ((it & 1) * MAXP + c) * 12 + 2 * row;
                unsigned long long a0 = 0, a1 = 0;
                int budget = 1 << 20;
                for (;;) {
                    if (mine) {
                        a0 = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const bool ok = !mine || ((unsigned)(a0 >> 32) == epoch && (unsigned)(a1 >> 32) == epoch);
                    if (__all(ok)) break;
                    if (--budget == 0) {
                        if (lane == 0) atomicExch(bad, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                double d = mine ? __hiloint2double((int)(unsigned)a1, (int)(unsigned)a0) : 0.0;
                for (int o = 1; o < 16; o <<= 1) d += __shfl_xor(d, o);
                tot = d;
                // expected: sum over members (m+1)*(row+1) + it
                const double want = (double)(row + 1) * P * (P + 1) / 2 + (double)it * P;
                if (tot != want) atomicExch(bad, 2);
            }
        } else {
            const int nline = (MODE == 3) ? 2 : 1;
            if (wave == 0) {
                const long long val = ((long long)(member + 1) * (lane + 1) + it) * 1024;   // fixed point
                unsigned long long* lp;
                if (MODE == 3) lp = base + (it & 1) * 32 + (lane < 2 ? lane : 8 + (lane - 2));
                else lp = base + (it & 1) * 16 + lane;
                if (lane < 6) {
                    const unsigned long long add = ((unsigned long long)val << 5) + 1ull;
                    if (MODE == 2) __hip_atomic_fetch_add(lp, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_fetch_add(lp, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                unsigned long long x = 0;
                int budget = 1 << 20;
                const unsigned long long pv = prev[it & 1];
                for (;;) {
                    if (lane < 6) {
                        if (MODE == 4) x = __hip_atomic_fetch_or(lp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else if (MODE == 5) x = __builtin_nontemporal_load(lp);
                        else x = __hip_atomic_load(lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const bool ok = lane >= 6 || (((x - pv) & 31ull) == (unsigned long long)P);
                    if (__all(ok)) break;
                    if (--budget == 0) {
                        if (lane == 0) atomicExch(bad, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const long long diff = (long long)(x - pv) >> 5;
                prev[it & 1] = x;
                tot = (double)diff * (1.0 / 1024);
                const double want = (double)(lane + 1) * P * (P + 1) / 2 + (double)it * P;
                if (lane < 6 && tot != want) atomicExch(bad, 2);
                (void)nline;
            }
        }
        if (wave == 0 && lane == 0) s_v[0] = tot;
        const long long t1 = __builtin_amdgcn_s_memtime();
        t_total += t1 - t0;
        __syncthreads();
        acc += s_v[0] * 1e-30;
    }
    if (tid == 0) cyc[blockIdx.x] = t_total;
    sink[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static void run_xch(const char* name, int P, int work, long long stride_words = 1024) {
    const int n_ch = 8, rounds = 4000;
    unsigned long long* x;
    long long* c;
    double* sink;
    int* bad;
    (void)hipMalloc(&x, 8 * stride_words * n_ch + 4096);
    (void)hipMalloc(&c, 8 * 256);
    (void)hipMalloc(&sink, 8 * 256 * 256);
    (void)hipMalloc(&bad, 4);
    float best = 1e30f;
    long long hc[256];
    int hb = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(x, 0, 8 * stride_words * n_ch + 4096);
        (void)hipMemset(bad, 0, 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        xch_kernel<MODE><<<n_ch * P, 256>>>(x, c, P, rounds, work, sink, bad, stride_words);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        (void)hipMemcpy(hc, c, 8 * n_ch * P, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    }
    double mean = 0;
    for (int i = 0; i < n_ch * P; ++i) mean += (double)hc[i] / rounds;
    mean /= n_ch * P;
    printf("B  %-34s P %2d work %4d stride %8lld B: %7.3f us per round, exchange segment %7.1f cycles (mean)%s | per channel:", name, P,
           work, stride_words * 8, best * 1e3 / rounds, mean, hb ? "  ** CHECK FAILED **" : "");
    for (int chn = 0; chn < n_ch; ++chn) {
        double m2 = 0;
        for (int mm = 0; mm < P; ++mm) {
            // block index of (channel, member): bq = (chn / 8) * P + mm, br = chn % 8
            m2 += (double)hc[((chn / 8) * P + mm) * 8 + (chn % 8)] / rounds;
        }
        printf(" %5.0f", m2 / P);
    }
    printf("\n");
    (void)hipFree(x);
    (void)hipFree(c);
    (void)hipFree(sink);
    (void)hipFree(bad);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run_chain<0>("fma_f64", 1.0000001, 1e-9);
    run_chain<1>("add_f64", 1.0000001, 1e-9);
    run_chain<2>("v_rcp_f64 + add", 1.3, 1e-9);
    run_chain<3>("v_rsq_f64 + add", 1.3, 1e-9);
    run_chain<4>("v_sqrt_f64 + add", 1.3, 1e-9);
    run_chain<5>("sqrt() + add", 1.3, 1e-9);
    run_chain<6>("b / v + add (IEEE division)", 1.3, 0.7);
    run_chain<7>("atan(b / v) + add (libm)", 1.3, 0.7);
    run_chain<15>("atan(v) + add (libm)", 0.3, 0.7);
    run_chain<8>("sincos_turns + 3 ops", 0.3, 0.7);
    run_chain<9>("div_rn (5 ops) + add", 1.3, 0.7);
    run_chain<10>("mul, ceil, mul, add", 1.3, 0.7);
    run_chain<13>("cvt f64->i32, add, cvt i32->f64, mul", 100.3, 0.7);
    run_chain<14>("2 readlane + add", 1.3, 0.7);
    run_chain<11>("LDS store -> load (same wave) + add", 1.3, 0.7);
    run_chain<12>("LDS store -> __syncthreads -> load + add", 1.3, 0.7);
    for (long long stride : {64ll, 512ll, 8192ll, 131072ll, 1048576ll + 24}) {
        run_xch<0>("granules (plain store, sc1 poll)", 10, 300, stride);
        run_xch<1>("int64 atomics wg, sc1 poll", 10, 300, stride);
        run_xch<4>("int64 atomics wg, atomic-or poll", 10, 300, stride);
        run_xch<5>("int64 atomics wg, nt poll", 10, 300, stride);
        run_xch<2>("int64 atomics agent, sc1 poll", 10, 300, stride);
    }
    return 0;
}
