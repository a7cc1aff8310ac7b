// Throughput-mode tracking kernel for gfx950: one 256-thread workgroup per channel, two workgroups per CU,
// used when there are more channels than CUs (split == 1).  Same arithmetic contract as sgx_trk.hip
// (reference tracking.py:13-295) but a different map, chosen for instruction count
// rather than latency - in this regime the chip is bound by fp64 VALU issue, not by the dependency chain:
//
//   ONE LANE PER PROMPT CHIP.  Lane c takes the samples whose prompt index ceil(tP) equals c: [s0, s1), both
//   ends found with the exact reference arithmetic (estimate from 1/step plus two exact probes).  The prompt
//   code is constant there, and the early and late ramps (spanning < 1 chip) switch at most once each, at
//   eE and eL (normally the same sample, mid-chip).  So the lane's ~37 samples are a HEAD run [s0, e1) and a
//   TAIL run [e2, s1) with all three codes constant in each, so a run needs sum_k x_k B_k only (B_k: the phasor of
//   sample k of a run, a per-block table), rotated by the run-start phasor (four small tables + one table lookup),
//   the code signs applied once per lane.
//   THE SAMPLES ARE NEVER CONVERTED (round 3).  B_k 2^30 is rounded to an integer and written in four signed
//   radix-256 digits; digit l of four consecutive k is one dword, so v_dot4_i32_i8 of a dword of the record with it
//   adds four samples' products exactly: 8 dot products per 4 samples (4 digits x cos, sin) instead of 4 byte
//   extractions, 4 conversions and 8 fp64 FMAs, and the 40 weight dwords live in registers for the whole block (no
//   LDS read per sample).  The four int32 digit sums of a run are put together in fp64 (exact).  The table's
//   rounding (2^-31 per entry, random in sign) moves a block's sums by ~1e-6 of 1e5: 1e-11 relative.
//   Bytes are fetched at their byte address (the hardware takes unaligned loads); bytes past a run's end are masked
//   once per run.  A chip whose early and late switches differ (fp64 rounding exactly at a boundary) or whose runs
//   exceed 20 samples takes an exact per-sample loop.
//   OTHER SAMPLE TYPES (round 4; MODE 1 = uint8, 2 = little-endian int16).  The dot products take signed bytes, so a
//   record is split into planes of signed bytes and a constant: uint8 x = x' + 128 with x' = the byte with its top bit
//   flipped; int16 x = 256 h + (l' + 128) with h the high byte and l' the low byte with its top bit flipped (two v_perm
//   per pair of dwords de-interleave a run into a low and a high plane, four samples per dword like an int8 run, so the
//   same forty weight dwords serve).  sum_k x_k B_k = [256 sum h B] + sum x' B + 128 sum_{k < len} B_k; the last term from a
//   per-block table of prefix sums of the integer weights (exact).  int16: twice the dot products of int8.
#include "sgx_trk_common.h"
#include "sgx_trk_math.h"

// Threads per workgroup (one channel) and workgroups per CU the register budget is cut for.  Measured on 2048 channels
// x 500 ms: 256 x 2 13.8 ms (13.4 with the weight dwords kept in registers, which only this budget allows); 256 x 3 (168
// registers, spills) 15.3 ms; 128 x 4 (both waves busy in every phase) 15.4 ms.
#ifndef TP_THREADS
#define TP_THREADS 256
#endif
// workgroups per CU the register allocation is cut for, per sample type (MODE 0 int8, 1 uint8, 2 int16): -DTP_OCC2=1 gives
// the int16 instance - two planes of samples in flight - the registers of a whole SIMD lane
#ifndef TP_OCC2
#define TP_OCC2 2
#endif
#ifndef TP_OCC1
#define TP_OCC1 2
#endif
#define TP_OCC_MODE(M) ((M) == 2 ? TP_OCC2 : ((M) == 1 ? TP_OCC1 : TP_OCC))
#ifndef TP_OCC
#define TP_OCC 2
#endif

// workgroup barrier that waits for LDS traffic only: __syncthreads() also waits for the wave's global loads and stores
// (the record stores of a block, the look-ahead loads behind the last chip), whose round trips nobody here depends on
__device__ __forceinline__ void tp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// lane 15 of rows 0 and 2 to every lane of rows 1 and 3 (the other rows get 0)
__device__ __forceinline__ double tp_row_bcast15(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
    return __hiloint2double(ohi, olo);
}

typedef double tp_v2d __attribute__((ext_vector_type(2)));
typedef unsigned tp_v2u __attribute__((ext_vector_type(2)));

// first sample above thr from real arithmetic; `near` is raised when a sample lies within 1e-7 samples of the boundary
__device__ __forceinline__ int tp_bound(double start, double inv_step, double thr, bool& near) {
    const double u = (thr - start) * inv_step;
    const double f = floor(u);
    const double fr = u - f;
    near = near || !(fr > 1e-7 && fr < 1.0 - 1e-7);
    return (int)f + 1;
}

struct TpRamps {
    double startE, stepE, startP, stepP, startL, stepL, inv_step;
};
struct TpChip {
    int s0, s1, eE, eL;   // the chip's samples [s0, s1) and the early / late switch samples, clamped to s1
};

// Sample range of prompt chip c and the switch samples of the early and late ramps inside it.  Boundaries from one
// multiply each, u = (thr - start) / step in real arithmetic: the reference's ramp fl(fl(i step) + start) lies within
// 1e-11 samples of the real one, so floor(u) + 1 is the first sample above thr unless u is within 1e-7 of an integer -
// then (any lane of the wave) the exact probes decide.
__device__ __forceinline__ TpChip tp_chip_bounds(const TpRamps& R, int c, int c_first, int c_last, int blk) {
    bool near = false;
    int s0 = (c == c_first) ? 0 : tp_bound(R.startP, R.inv_step, (double)(c - 1), near);
    int s1 = (c == c_last) ? blk : tp_bound(R.startP, R.inv_step, (double)c, near);
    if (__builtin_expect(__any(near), 0)) {
        s0 = (c == c_first) ? 0 : first_above(R.startP, R.stepP, R.inv_step, (double)(c - 1));
        s1 = (c == c_last) ? blk : first_above(R.startP, R.stepP, R.inv_step, (double)c);
    }
    s0 = s0 < 0 ? 0 : s0;
    s1 = s1 > blk ? blk : s1;
    const int kE = (int)ceil(ramp_at(s0, R.stepE, R.startE));
    const int kL = (int)ceil(ramp_at(s0, R.stepL, R.startL));
    near = false;
    int eE = tp_bound(R.startE, R.inv_step, (double)kE, near);
    int eL = tp_bound(R.startL, R.inv_step, (double)kL, near);
    if (__builtin_expect(__any(near), 0)) {
        eE = first_above(R.startE, R.stepE, R.inv_step, (double)kE);
        eL = first_above(R.startL, R.stepL, R.inv_step, (double)kL);
    }
    TpChip o;
    o.s0 = s0;
    o.s1 = s1;
    o.eE = eE > s1 ? s1 : eE;
    o.eL = eL > s1 ? s1 : eL;
    return o;
}

// The same for a chip in the interior of the block when dllCorrelatorSpacing lies in [step, 1 - step]: at the chip's first
// sample the early ramp's index is c - 1 and the late ramp's is c (tP in (c - 1, c - 1 + step] there), so all four
// boundaries are floors of (thr - start) / step for known thresholds - no ramp evaluation, one guard for the four.
// `ok` is cleared when a boundary lies within 1e-7 samples of a sample (the wave then takes tp_chip_bounds).
__device__ __forceinline__ TpChip tp_chip_bounds_inner(const TpRamps& R, int c, int blk, bool& ok) {
    const double cm1 = (double)(c - 1);
    const double u0 = (cm1 - R.startP) * R.inv_step;
    const double u1 = u0 + R.inv_step;                         // (c - startP) / step to 1e-12 samples
    const double uE = (cm1 - R.startE) * R.inv_step;
    const double uL = ((cm1 + 1.0) - R.startL) * R.inv_step;
    const double f0 = floor(u0), f1 = floor(u1), fE = floor(uE), fL = floor(uL);
    const double d0 = fabs((u0 - f0) - 0.5), d1 = fabs((u1 - f1) - 0.5), dE = fabs((uE - fE) - 0.5), dL = fabs((uL - fL) - 0.5);
    ok = ok && fmax(fmax(d0, d1), fmax(dE, dL)) < 0.5 - 1e-7;
    TpChip o;
    const int s0 = (int)f0 + 1, s1 = (int)f1 + 1;
    o.s0 = s0 < 0 ? 0 : (s0 > blk ? blk : s0);              // (never outside the block, whatever the parameters)
    o.s1 = s1 < 0 ? 0 : (s1 > blk ? blk : s1);
    const int eE = (int)fE + 1, eL = (int)fL + 1;
    o.eE = eE > o.s1 ? o.s1 : eE;
    o.eL = eL > o.s1 ? o.s1 : eL;
    return o;
}

// 20 samples starting at sample `samp` of the channel's grid (any byte alignment; the hardware takes unaligned 16-byte
// loads): five dwords of one-byte samples, ten of int16
template <int MODE>
__device__ __forceinline__ void tp_load_raw(const int8_t* __restrict__ rec, long long samp, unsigned (&w)[MODE == 2 ? 10 : 5]) {
    if constexpr (MODE == 2) {
        const long long addr = samp * 2;
        const U4a q = *reinterpret_cast<const U4a*>(rec + addr);
        const U4a r = *reinterpret_cast<const U4a*>(rec + addr + 16);
        const U2a t = *reinterpret_cast<const U2a*>(rec + addr + 32);
        w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
        w[4] = r.x, w[5] = r.y, w[6] = r.z, w[7] = r.w;
        w[8] = t.x, w[9] = t.y;
    } else {
        const U4a q = *reinterpret_cast<const U4a*>(rec + samp);
        const unsigned q4 = reinterpret_cast<const U2a*>(rec + samp + 16)->x;
        w[0] = q.x;
        w[1] = q.y;
        w[2] = q.z;
        w[3] = q.w;
        w[4] = q4;
    }
}

// the planes of signed bytes of a run (see the header): lo for every type, hi for int16
template <int MODE>
__device__ __forceinline__ void tp_planes(const unsigned (&w)[MODE == 2 ? 10 : 5], unsigned (&lo)[5], unsigned (&hi)[5]) {
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        if constexpr (MODE == 0) {
            lo[d] = w[d];
            hi[d] = 0u;
        } else if constexpr (MODE == 1) {
            lo[d] = w[d] ^ 0x80808080u;
            hi[d] = 0u;
        } else {
            lo[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x06040200u) ^ 0x80808080u;
            hi[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x07050301u);
        }
    }
}

// one sample of the channel's grid as the reference's float64 arithmetic sees it (the exact per-sample path)
template <int MODE>
__device__ __forceinline__ double tp_sample(const int8_t* __restrict__ rec, long long samp, long long alloc) {
    if constexpr (MODE == 2) {
        long long a = samp * 2;
        a = a > alloc - 2 ? alloc - 2 : a;
        return (double)reinterpret_cast<const AnyAt<short>*>(rec + a)->v;
    } else {
        const long long a = samp > alloc - 1 ? alloc - 1 : samp;
        return MODE == 1 ? (double)(int)(unsigned char)rec[a] : (double)(int)rec[a];
    }
}

// bytes >= len zeroed
__device__ __forceinline__ void tp_mask_run(int len, unsigned (&w)[5]) {
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        int keep = len - 4 * d;                      // bytes of this dword inside the run
        keep = keep < 0 ? 0 : (keep > 4 ? 4 : keep);
        w[d] &= (keep >= 4) ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u);
    }
}

// B_k 2^30 (k < 20) as four signed radix-256 digits, most significant first: lane = 20 comp + k
// (returns the lane's integer weight W_k = rint(B_k 2^30), 0 for lanes >= 40)
__device__ __forceinline__ int tp_weight_digits(const TpCarr& t, signed char (&wq)[2][4][32], int lane) {
    int w_int = 0;
    if (lane < 40) {
        const int comp = lane / 20, k = lane - 20 * comp;
        const double v = comp ? t.B[k].y : t.B[k].x;
        int b = (int)rint(v * 1073741824.0);
        w_int = b;
        const int d3 = (int)(signed char)(b & 0xFF);
        b = (b - d3) >> 8;
        const int d2 = (int)(signed char)(b & 0xFF);
        b = (b - d2) >> 8;
        const int d1 = (int)(signed char)(b & 0xFF);
        b = (b - d1) >> 8;
        wq[comp][0][k] = (signed char)b;
        wq[comp][1][k] = (signed char)d1;
        wq[comp][2][k] = (signed char)d2;
        wq[comp][3][k] = (signed char)d3;
    }
    return w_int;
}

// pref[j] = sum_{k < j} (W_k cos, W_k sin), j = 0..20: the constant of a run of j unsigned bytes (integers below 2^35:
// every addition exact).  A scan over lanes 0..19 (cos) and 20..39 (sin).
__device__ __forceinline__ void tp_weight_prefix(int w_int, double2 (&pref)[24], int lane) {
    const int comp = lane >= 20 ? 1 : 0, k = lane - 20 * comp;
    double v = lane < 40 ? (double)w_int : 0.0;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const double o = __shfl_up(v, off);
        if (k >= off) v += o;
    }
    if (lane < 40) reinterpret_cast<double*>(&pref[k + 1])[comp] = v;
    if (lane == 0) pref[0] = make_double2(0.0, 0.0);
}

// the four digit sums of a run -> sum_k x_k B_k 2^30 (exact in fp64: < 2^45)
__device__ __forceinline__ double tp_join(int a0, int a1, int a2, int a3) {
    const int hi = a0 * 256 + a1, lo = a2 * 256 + a3;       // |a_l| < 20 * 127 * 128: both fit 32 bits
    return __builtin_fma((double)hi, 65536.0, (double)lo);
}

__device__ __forceinline__ int tp_dot4_first(int x, int w) {
    int d;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(d) : "v"(x), "v"(w));
    return d;
}

// Head and tail run of a chip (bytes beyond the runs already zero): sum_k x_k B_k of each by int8 dot products against
// the digit dwords, the runs rotated by their start phasors, the three codes applied.
__device__ __forceinline__ void tp_dots(const int (&wr)[2][4][5], const unsigned (&wh)[5], const unsigned (&wt)[5],
                                        double& Hc, double& Hs, double& Tc, double& Ts) {
    // weight dwords [cos, sin][digit][d]: dword d holds the digits of k = 4 d .. 4 d + 3; in registers for the whole block
    int hc[4], hs[4], tc[4], ts[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int4 wc = make_int4(wr[0][l][0], wr[0][l][1], wr[0][l][2], wr[0][l][3]);
        const int4 ws = make_int4(wr[1][l][0], wr[1][l][1], wr[1][l][2], wr[1][l][3]);
        const int wc4 = wr[0][l][4], ws4 = wr[1][l][4];
        // (the three-address form with a literal zero: the accumulating v_dot4c would want sixteen zeroed registers)
        int a = tp_dot4_first((int)wh[0], wc.x);
        int b = tp_dot4_first((int)wh[0], ws.x);
        int e = tp_dot4_first((int)wt[0], wc.x);
        int f = tp_dot4_first((int)wt[0], ws.x);
        a = __builtin_amdgcn_sdot4((int)wh[1], wc.y, a, false);
        b = __builtin_amdgcn_sdot4((int)wh[1], ws.y, b, false);
        e = __builtin_amdgcn_sdot4((int)wt[1], wc.y, e, false);
        f = __builtin_amdgcn_sdot4((int)wt[1], ws.y, f, false);
        a = __builtin_amdgcn_sdot4((int)wh[2], wc.z, a, false);
        b = __builtin_amdgcn_sdot4((int)wh[2], ws.z, b, false);
        e = __builtin_amdgcn_sdot4((int)wt[2], wc.z, e, false);
        f = __builtin_amdgcn_sdot4((int)wt[2], ws.z, f, false);
        a = __builtin_amdgcn_sdot4((int)wh[3], wc.w, a, false);
        b = __builtin_amdgcn_sdot4((int)wh[3], ws.w, b, false);
        e = __builtin_amdgcn_sdot4((int)wt[3], wc.w, e, false);
        f = __builtin_amdgcn_sdot4((int)wt[3], ws.w, f, false);
        hc[l] = __builtin_amdgcn_sdot4((int)wh[4], wc4, a, false);
        hs[l] = __builtin_amdgcn_sdot4((int)wh[4], ws4, b, false);
        tc[l] = __builtin_amdgcn_sdot4((int)wt[4], wc4, e, false);
        ts[l] = __builtin_amdgcn_sdot4((int)wt[4], ws4, f, false);
    }
    Hc = tp_join(hc[0], hc[1], hc[2], hc[3]), Hs = tp_join(hs[0], hs[1], hs[2], hs[3]);
    Tc = tp_join(tc[0], tc[1], tc[2], tc[3]), Ts = tp_join(ts[0], ts[1], ts[2], ts[3]);
}

// the runs' sums (scaled by 2^30: the 2^-30 rides on gh) rotated by their start phasors, the three codes applied
__device__ __forceinline__ void tp_apply(double2 gh, double2 gt, double Hc, double Hs, double Tc, double Ts,
                                         bool e_switched, bool l_switched, double cP, double cEh, double cEn, double cLh,
                                         double cLn, double& aIE, double& aQE, double& aIP, double& aQP, double& aIL,
                                         double& aQL) {
    // rotate the runs by their start phasors: cos part -> Q, sin part -> I (tracking.py:205-207)
    const double hQ = __builtin_fma(gh.x, Hc, -(gh.y * Hs)), hI = __builtin_fma(gh.y, Hc, gh.x * Hs);
    const double tQ = __builtin_fma(gt.x, Tc, -(gt.y * Ts)), tI = __builtin_fma(gt.y, Tc, gt.x * Ts);
    // code of the tail: switched iff the ramp's switch sample is the run boundary
    const double cEt = e_switched ? cEn : cEh;
    const double cLt = l_switched ? cLn : cLh;
    aIE = __builtin_fma(cEt, tI, __builtin_fma(cEh, hI, aIE));
    aQE = __builtin_fma(cEt, tQ, __builtin_fma(cEh, hQ, aQE));
    aIP = __builtin_fma(cP, tI + hI, aIP);
    aQP = __builtin_fma(cP, tQ + hQ, aQP);
    aIL = __builtin_fma(cLt, tI, __builtin_fma(cLh, hI, aIL));
    aQL = __builtin_fma(cLt, tQ, __builtin_fma(cLh, hQ, aQL));
}

template <int MODE>
__global__ __launch_bounds__(TP_THREADS, TP_OCC_MODE(MODE)) void trk_kernel_tp(const int8_t* __restrict__ rec0,
                                                                const int8_t* __restrict__ codes,
                                                                const TrkChan* __restrict__ chans,
                                                                double* __restrict__ out, int* __restrict__ ms_done,
                                                                TrkConst K) {
    __shared__ unsigned s_code_hi[1028];   // hi dword of +-1.0 for [c1022, c0..c1022, c0] (tracking.py:111)
    __shared__ TrkBlock s_blk;             // code part used; carrier part unused here
    __shared__ TpCarr s_car;
    __shared__ __attribute__((aligned(16))) signed char s_wq[2][4][32];   // [cos, sin][digit, most significant first][k]: B_k 2^30 in radix 256
    __shared__ double s_red[6][TP_THREADS];
    __shared__ double s_tot[6];
    __shared__ TrkState s_st;
    __shared__ double s_rec[16][16];       // the records of the last sixteen blocks, [block & 15][series]: stored eight blocks
                                           // at a time, 64 contiguous bytes per series row (one 8-byte store per row and
                                           // block made 5.2 bytes of memory traffic per byte of series)
    __shared__ double2 s_pref[24];         // MODE != 0: prefix sums of the integer weights (tp_weight_prefix)
    constexpr int SB = MODE == 2 ? 2 : 1;  // bytes per sample
    constexpr int NW = 5 * SB;             // dwords of a run as loaded

    const int ch = blockIdx.x;
    if (ch >= K.n_ch) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const TrkChan cc = chans[ch];
    if (cc.prn == 0) {
        if (tid == 0) ms_done[ch] = 0;
        return;
    }
    // the channel's sample grid starts at byte cc.pad of the record (int16 channels may start on an odd byte,
    // tracking.py:107); K.rec_len / K.rec_alloc are bytes
    const int8_t* __restrict__ const rec = rec0 + cc.pad;
    const long long rec_samples = (K.rec_len - cc.pad) / SB;
    const long long alloc_bytes = K.rec_alloc - cc.pad;
    for (int i = tid; i < 1028; i += TP_THREADS) {
        int j = i - 1;
        if (j < 0) j = 1022;
        if (j >= 1023) j -= 1023;
        if (j >= 1023) j -= 1023;
        s_code_hi[i] = (codes[(cc.prn - 1) * 1023 + j] > 0) ? 0x3FF00000u : 0xBFF00000u;
    }
    if (tid == 0) {   // tracking.py:114-130
        s_st.codeFreq = K.code_basis;
        s_st.remCode = 0.0;
        s_st.oldCodeNco = s_st.oldCodeErr = 0.0;
        s_st.pos = cc.pos0;
        s_st.carrFreq = cc.acquiredFreq;
        s_st.carrBasis = cc.acquiredFreq;
        s_st.remCarr = 0.0;
        s_st.w = (cc.acquiredFreq * 2.0) * M_PI;
        s_st.oldCarrNco = s_st.oldCarrErr = 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        tp_tables(K, s_st.w, s_st.remCarr, s_car, lane, 0);
        tp_tables(K, s_st.w, s_st.remCarr, s_car, lane, 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int w_int = tp_weight_digits(s_car, s_wq, lane);
        if constexpr (MODE != 0) tp_weight_prefix(w_int, s_pref, lane);
    }
    if (wave == 1) prep_code(K, s_st.codeFreq, s_st.remCode, s_st.pos, s_st, s_blk, lane == 0, rec_samples);
    __syncthreads();

    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    const long long m = K.ms;
    const double two_pi = 2 * M_PI;
    int done = 0;
#ifdef TP_PROF
    // (diagnosis build: cycles per phase of a block, printed by channels 0 and 1000; tools/build_variant.sh ... -DTP_PROF)
    long long tp_acc[6] = {0, 0, 0, 0, 0, 0};
#define TP_STAMP(i) const long long tp_t##i = (long long)__builtin_amdgcn_s_memtime();
#else
#define TP_STAMP(i)
#endif
    for (int it = 0; it < K.ms; ++it) {
        TP_STAMP(0)
        const long long pos = s_blk.pos;
        const int blk = s_blk.blk;
        if (s_blk.stop) break;   // short read: tracking.py:159-163
        const double startE = s_blk.startE, stepE = s_blk.stepE;
        const double startP = s_blk.startP, stepP = s_blk.stepP;
        const double startL = s_blk.startL, stepL = s_blk.stepL;
        const double inv_step = s_blk.inv_step;
        const int c_first = (int)ceil(ramp_at(0, stepP, startP));
        const int c_last = (int)ceil(ramp_at(blk - 1, stepP, startP));

        double aIE = 0.0, aQE = 0.0, aIP = 0.0, aQP = 0.0, aIL = 0.0, aQL = 0.0;
        // A lane takes chips c_first + tid, + 256, ... (four or five per block).  The bytes of the NEXT chip are requested
        // before the current one is worked on, so their latency hides behind ~200 instructions instead of stalling the
        // wave in front of every chip.  (A lane without a next chip computes one for chip c_last: nothing is read beyond
        // the block, no lane-dependent branch.)
        const TpRamps R = {startE, stepE, startP, stepP, startL, stepL, inv_step};
        // interior chips take the short boundary computation when the spacing allows it (see tp_chip_bounds_inner)
        const bool inner_ok = (K.spacing >= stepP + 1e-6) && (K.spacing <= 1.0 - stepP - 1e-6);
        auto bounds_of = [&](int cc_) -> TpChip {
            bool ok = inner_ok && cc_ != c_first && cc_ != c_last;
            TpChip o = tp_chip_bounds_inner(R, cc_, blk, ok);
            if (__builtin_expect(!__all(ok), 0)) o = tp_chip_bounds(R, cc_, c_first, c_last, blk);
            return o;
        };
        int c = c_first + tid;
        bool have = c <= c_last;
        TpChip cur = bounds_of(have ? c : c_last);
        unsigned wh[NW], wt[NW];
        tp_load_raw<MODE>(rec, pos + cur.s0, wh);
        tp_load_raw<MODE>(rec, pos + (cur.eE > cur.eL ? cur.eE : cur.eL), wt);
        // the block's 40 weight dwords (two workgroups per CU leave 256 registers per lane: no LDS read per chip - with
        // them re-read for every chip the kernel measured 4 % slower)
        int wr[2][4][5];
        {
            const int* w32 = reinterpret_cast<const int*>(&s_wq[0][0][0]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int l = 0; l < 4; ++l)
#pragma unroll
                    for (int d = 0; d < 5; ++d) wr[q][l][d] = w32[(q * 4 + l) * 8 + d];
        }
        // one chip: `ck` with its bytes in wh / wt (the runs, unmasked), while the next chip's bounds go to `nk` and its
        // bytes are requested into yh / yt.  Called alternately with the two register sets swapped, so nothing is copied.
        auto chip = [&](const TpChip& ck, int cc_, unsigned (&wh)[NW], unsigned (&wt)[NW], TpChip& nk, int cn_, unsigned (&yh)[NW],
                        unsigned (&yt)[NW]) {
            const int s0 = ck.s0, s1 = ck.s1, eE = ck.eE, eL = ck.eL;
            const int e1 = eE < eL ? eE : eL, e2 = eE < eL ? eL : eE;
            const int len_h = e1 - s0, len_t = s1 - e2;
            const int kE = (int)ceil(ramp_at(s0, stepE, startE));
            const int kL = (int)ceil(ramp_at(s0, stepL, startL));
            // everything this chip reads from LDS - five code signs, the four phasor tables' entries, the tail's offset
            // phasor - requested HERE, in front of the next chip's boundary arithmetic, and waited for once (left to the
            // compiler the reads sit next to their uses in three batches, a wait in front of each: 18.86 -> 18.69 ms)
            unsigned hP;
            tp_v2u hE, hL;
            tp_v2d t3, t2, t1, t0, tb;
            {
                const unsigned ac = (unsigned)(unsigned long long)&s_code_hi[0];
                const unsigned at = (unsigned)(unsigned long long)&s_car;
                const unsigned aP = ac + 4u * (unsigned)cc_, aE = ac + 4u * (unsigned)kE, aL = ac + 4u * (unsigned)kL;
                const unsigned a3 = at + (unsigned)offsetof(TpCarr, W3) + 16u * (unsigned)(s0 >> 12);
                const unsigned a2 = at + (unsigned)offsetof(TpCarr, W2) + 16u * (unsigned)((s0 >> 8) & 15);
                const unsigned a1 = at + (unsigned)offsetof(TpCarr, W1) + 16u * (unsigned)((s0 >> 4) & 15);
                const unsigned a0 = at + (unsigned)offsetof(TpCarr, B) + 16u * (unsigned)(s0 & 15);
                const unsigned ab = at + (unsigned)offsetof(TpCarr, B) + 16u * (unsigned)((e2 - s0) & 31);
                asm volatile("ds_read_b32 %0, %1" : "=v"(hP) : "v"(aP));
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(hE) : "v"(aE));
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(hL) : "v"(aL));
                asm volatile("ds_read_b128 %0, %1" : "=v"(t3) : "v"(a3));
                asm volatile("ds_read_b128 %0, %1" : "=v"(t2) : "v"(a2));
                asm volatile("ds_read_b128 %0, %1" : "=v"(t1) : "v"(a1));
                asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1" : "=v"(tb) : "v"(ab));
            }
            nk = bounds_of(cn_);
            tp_load_raw<MODE>(rec, pos + nk.s0, yh);
            tp_load_raw<MODE>(rec, pos + (nk.eE > nk.eL ? nk.eE : nk.eL), yt);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hP), "+v"(hE), "+v"(hL), "+v"(t3), "+v"(t2), "+v"(t1), "+v"(t0), "+v"(tb));
            const double cP = __hiloint2double((int)hP, 0);
            const double cEh = __hiloint2double((int)hE.x, 0), cEn = __hiloint2double((int)hE.y, 0);
            const double cLh = __hiloint2double((int)hL.x, 0), cLn = __hiloint2double((int)hL.y, 0);
            const double2 gh = cmul2(cmul2(make_double2(t3.x, t3.y), make_double2(t2.x, t2.y)),
                                     cmul2(make_double2(t1.x, t1.y), make_double2(t0.x, t0.y)));
            const double2 b_tail = make_double2(tb.x, tb.y);
            const double q30 = 9.3132257461547852e-10;   // 2^-30: the scale of the weight digits
            const double2 ghq = make_double2(gh.x * q30, gh.y * q30);
            // the usual chip: early and late switch at the same sample, both runs 17..20 samples long: only the fifth
            // dword of a run has bytes to mask
            const bool usual = (e1 == e2) && ((unsigned)(len_h - 17) <= 3u) && ((unsigned)(len_t - 17) <= 3u);
            bool by_runs = true;
            // the exact per-sample loop over a chip (rare)
            auto per_sample = [&]() {
                if (s1 > s0) {
                    double2 ph = gh;
                    const double2 b1 = s_car.B[1];
                    for (int i = s0; i < s1; ++i) {
                        const double xd = tp_sample<MODE>(rec, pos + i, alloc_bytes);
                        const double xs = ph.y * xd, xc = ph.x * xd;
                        const double cE = i >= eE ? cEn : cEh;
                        const double cL = i >= eL ? cLn : cLh;
                        aIE = __builtin_fma(cE, xs, aIE);
                        aQE = __builtin_fma(cE, xc, aQE);
                        aIP = __builtin_fma(cP, xs, aIP);
                        aQP = __builtin_fma(cP, xc, aQP);
                        aIL = __builtin_fma(cL, xs, aIL);
                        aQL = __builtin_fma(cL, xc, aQL);
                        ph = cmul2(ph, b1);
                    }
                }
            };
            if constexpr (MODE == 0) {
                // int8: the record's dwords ARE the plane (masked in place)
                if (__builtin_expect(__all(usual), 1)) {
                    wh[4] &= 0xFFFFFFFFu >> (8 * (20 - len_h));
                    wt[4] &= 0xFFFFFFFFu >> (8 * (20 - len_t));
                } else {
                    const bool odd = (s1 > s0) && (e2 != e1 || len_h > TP_RUN || len_t > TP_RUN || e2 - s0 > 31);
                    if (__any(odd)) {
                        by_runs = false;
                        per_sample();
                    } else {
                        // short or empty runs (the block's first and last chip): every dword masked (a lane without
                        // samples masks everything)
                        tp_mask_run(s1 > s0 ? len_h : 0, wh);
                        tp_mask_run(s1 > s0 ? len_t : 0, wt);
                    }
                }
                if (by_runs) {
                    const double2 gt = cmul2(ghq, b_tail);
                    double Hc, Hs, Tc, Ts;
                    tp_dots(wr, wh, wt, Hc, Hs, Tc, Ts);
                    tp_apply(ghq, gt, Hc, Hs, Tc, Ts, eE <= e2, eL <= e2, cP, cEh, cEn, cLh, cLn, aIE, aQE, aIP, aQP, aIL, aQL);
                }
            } else {
                // planes of signed bytes: four samples per dword whatever the sample type
                unsigned ph_lo[5], ph_hi[5], pt_lo[5], pt_hi[5];
                tp_planes<MODE>(wh, ph_lo, ph_hi);
                tp_planes<MODE>(wt, pt_lo, pt_hi);
                int ln_h = len_h, ln_t = len_t;        // run lengths as the constant term counts them
                if (__builtin_expect(__all(usual), 1)) {
                    ph_lo[4] &= 0xFFFFFFFFu >> (8 * (20 - len_h));
                    pt_lo[4] &= 0xFFFFFFFFu >> (8 * (20 - len_t));
                    if constexpr (MODE == 2) {
                        ph_hi[4] &= 0xFFFFFFFFu >> (8 * (20 - len_h));
                        pt_hi[4] &= 0xFFFFFFFFu >> (8 * (20 - len_t));
                    }
                } else {
                    const bool odd = (s1 > s0) && (e2 != e1 || len_h > TP_RUN || len_t > TP_RUN || e2 - s0 > 31);
                    if (__any(odd)) {
                        by_runs = false;
                        per_sample();
                    } else {
                        ln_h = s1 > s0 ? len_h : 0;
                        ln_t = s1 > s0 ? len_t : 0;
                        tp_mask_run(ln_h, ph_lo);
                        tp_mask_run(ln_t, pt_lo);
                        if constexpr (MODE == 2) {
                            tp_mask_run(ln_h, ph_hi);
                            tp_mask_run(ln_t, pt_hi);
                        }
                    }
                }
                if (by_runs) {
                    double Hc, Hs, Tc, Ts;
                    tp_dots(wr, ph_lo, pt_lo, Hc, Hs, Tc, Ts);
                    if constexpr (MODE == 2) {
                        double Gc, Gs, Uc, Us;
                        tp_dots(wr, ph_hi, pt_hi, Gc, Gs, Uc, Us);
                        Hc = __builtin_fma(Gc, 256.0, Hc), Hs = __builtin_fma(Gs, 256.0, Hs);
                        Tc = __builtin_fma(Uc, 256.0, Tc), Ts = __builtin_fma(Us, 256.0, Ts);
                    }
                    const double2 kh = s_pref[ln_h], kt = s_pref[ln_t];   // (all integers below 2^53: exact)
                    Hc = __builtin_fma(kh.x, 128.0, Hc), Hs = __builtin_fma(kh.y, 128.0, Hs);
                    Tc = __builtin_fma(kt.x, 128.0, Tc), Ts = __builtin_fma(kt.y, 128.0, Ts);
                    tp_apply(ghq, cmul2(ghq, b_tail), Hc, Hs, Tc, Ts, eE <= e2, eL <= e2, cP, cEh, cEn, cLh, cLn, aIE, aQE, aIP,
                             aQP, aIL, aQL);
                }
            }
        };
        TpChip nxt;
        unsigned nh[NW], nt[NW];
        TP_STAMP(1)
#pragma unroll 1
        while (have) {
            int cn = c + TP_THREADS;
            bool have_n = cn <= c_last;
            chip(cur, c, wh, wt, nxt, have_n ? cn : c, nh, nt);   // (no next chip: the current one again - an interior chip, no branch)
            c = cn;
            have = have_n;
            if (!have) break;
            cn = c + TP_THREADS;
            have_n = cn <= c_last;
            chip(nxt, c, nh, nt, cur, have_n ? cn : c, wh, wt);
            c = cn;
            have = have_n;
        }
        TP_STAMP(2)
        s_red[0][tid] = aIE;
        s_red[1][tid] = aQE;
        s_red[2][tid] = aIP;
        s_red[3][tid] = aQP;
        s_red[4][tid] = aIL;
        s_red[5][tid] = aQL;
        tp_barrier();
        TP_STAMP(3)
        if (tid < 192) {
            // sum v = tid / 32 is folded by two rows of 16 lanes: eight values per lane (one batch of reads, a tree of
            // adds), a DPP row sum, the first row's sum handed to the second.  Fixed order: deterministic.
            // (sixteen values per lane and one row per sum: 860 cycles of every block, most of it the chain of adds)
            static_assert(TP_THREADS == 256, "eight values per lane");
            const int v = tid >> 5, l = tid & 31;
            const double x0 = s_red[v][l], x1 = s_red[v][l + 32], x2 = s_red[v][l + 64], x3 = s_red[v][l + 96];
            const double x4 = s_red[v][l + 128], x5 = s_red[v][l + 160], x6 = s_red[v][l + 192], x7 = s_red[v][l + 224];
            double acc = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
            acc = row_sum(acc);
            acc += tp_row_bcast15(acc);          // rows 1 and 3 of a wave: their own sum plus the row's before
            if (l == 31) s_tot[v] = acc;
        }
        tp_barrier();
        TP_STAMP(4)
        const bool more = (it + 1 < K.ms);
        if (wave == 0) {
            // T7 PLL (tracking.py:223-235), T5 end-of-block carrier phase, tables of the next block
            const double I_P = s_tot[2], Q_P = s_tot[3];
            const double oldNco = s_st.oldCarrNco, oldErr = s_st.oldCarrErr, basis = s_st.carrBasis;
            const double arg_end = s_st.w * ((double)blk / K.fs) + s_st.remCarr;
            const double kq = floor(arg_end * K.inv_2pi);
            double rc = __builtin_fma(-kq, two_pi, arg_end);
            if (rc < 0.0) rc += two_pi;
            if (rc >= two_pi) rc -= two_pi;
            const double carrError = sgx_atan_ratio(Q_P, I_P) * K.inv_2pi;   // atan(Q/I) / 2 / pi to 1.5 ulp (sgx_trk_math.h)
            const double carrNco = oldNco + K.k_carr_a * (carrError - oldErr) + carrError * K.k_carr_b;
            const double carrFreq = basis + carrNco;
            const double w_new = (carrFreq * 2.0) * M_PI;
            if (more) {
                tp_tables(K, w_new, rc, s_car, lane, 0);
                tp_tables(K, w_new, rc, s_car, lane, 1);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                const int w_int = tp_weight_digits(s_car, s_wq, lane);
                if constexpr (MODE != 0) tp_weight_prefix(w_int, s_pref, lane);
            }
            if (lane == 0) {
                s_st.w = w_new;
                s_st.remCarr = rc;
                s_st.oldCarrNco = carrNco;
                s_st.oldCarrErr = carrError;
                s_st.carrFreq = carrFreq;
                double* __restrict__ r = s_rec[it & 15];  // T9 record (tracking.py:255-275): stored by wave 3, up to eight blocks later
                r[2] = carrFreq;
                r[3] = I_P;
                r[4] = s_tot[0];
                r[5] = s_tot[4];
                r[6] = s_tot[1];
                r[7] = Q_P;
                r[8] = s_tot[5];
                r[11] = carrError;
                r[12] = carrNco;
            }
        } else if (wave == 1) {
            // T8 DLL (tracking.py:238-251), then block size and ramps of the next block (T1, T3, T4)
            const double I_E = s_tot[0], Q_E = s_tot[1], I_L = s_tot[4], Q_L = s_tot[5];
            const double oldNco = s_st.oldCodeNco, oldErr = s_st.oldCodeErr;
            const long long pos_after = s_st.pos;
            const double rem_next = s_st.remCode;
            const double eE = sgx_sqrt1(I_E * I_E + Q_E * Q_E);      // (<= 1 ulp each: sgx_trk_math.h)
            const double eL = sgx_sqrt1(I_L * I_L + Q_L * Q_L);
            const double codeError = sgx_div1(eE - eL, eE + eL);
            const double codeNco = oldNco + K.k_code_a * (codeError - oldErr) + codeError * K.k_code_b;
            const double codeFreq = K.code_basis - codeNco;
            if (lane == 0) {
                s_st.oldCodeNco = codeNco;
                s_st.oldCodeErr = codeError;
                s_st.codeFreq = codeFreq;
                double* __restrict__ r = s_rec[it & 15];
                r[0] = (double)(pos_after * SB + cc.pad + K.file_off);   // fid.tell(): bytes (tracking.py:255)
                r[1] = codeFreq;
                r[9] = codeError;
                r[10] = codeNco;
            }
            if (more) prep_code(K, codeFreq, rem_next, pos_after, s_st, s_blk, lane == 0, rec_samples);
        } else if (wave == TP_THREADS / 64 - 1) {
            // the records of the eight blocks before this one: written to LDS by the filter waves, stored by this wave
            // while it has nothing else to do - lane = 8 row + j stores block it - 8 + j of series row `row`, so the eight
            // lanes of a row write 64 contiguous bytes (on the filter waves the thirteen stores and their addresses were
            // 800 cycles of the block's critical path)
            if (it > 0 && (it & 7) == 0) {
                const int j = lane & 7, k = it - 8 + j;
                o[(lane >> 3) * m + k] = s_rec[k & 15][lane >> 3];
                if (lane < 8 * (SGX_NUM_SERIES - 8)) o[(8 + (lane >> 3)) * m + k] = s_rec[k & 15][8 + (lane >> 3)];
            }
        }
        done = it + 1;
        TP_STAMP(5)
        tp_barrier();   // next block's parameters visible
#ifdef TP_PROF
        {
            const long long tp_t6 = (long long)__builtin_amdgcn_s_memtime();
            tp_acc[0] += tp_t1 - tp_t0;
            tp_acc[1] += tp_t2 - tp_t1;
            tp_acc[2] += tp_t3 - tp_t2;
            tp_acc[3] += tp_t4 - tp_t3;
            tp_acc[4] += tp_t5 - tp_t4;
            tp_acc[5] += tp_t6 - tp_t5;
        }
#endif
    }
#ifdef TP_PROF
    if ((ch == 0 || ch == 1000) && lane == 0 && done > 0)
        printf("[tp prof] ch %d wave %d cycles/block: top %lld loop %lld red-barrier %lld fold %lld filter %lld end-barrier %lld\n",
               ch, wave, tp_acc[0] / done, tp_acc[1] / done, tp_acc[2] / done, tp_acc[3] / done, tp_acc[4] / done, tp_acc[5] / done);
#endif
    // the records not stored yet: blocks 8 floor((done - 1) / 8) .. done - 1 (the loop's final barrier has made them visible)
    if (done > 0 && tid < 8 * SGX_NUM_SERIES) {
        const int k = ((done - 1) & ~7) + (tid & 7);
        if (k < done) o[(tid >> 3) * m + k] = s_rec[k & 15][tid >> 3];
    }
    if (tid == 0) ms_done[ch] = done;
}

void sgx_trk_tp_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                       double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                       int* err) {
    (void)prof;
    (void)xch;
    (void)err;
    if (K.kind == SGX_DT_INT16) trk_kernel_tp<2><<<K.n_ch, TP_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K);
    else if (K.kind == SGX_DT_UINT8) trk_kernel_tp<1><<<K.n_ch, TP_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K);
    else trk_kernel_tp<0><<<K.n_ch, TP_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K);
    (void)n_blocks;
}
