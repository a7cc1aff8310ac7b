// TrackingResult.track on gfx950 (reference tracking.py:13-295).
//
// One persistent 512-thread workgroup (8 waves, two per SIMD of one CU) per channel walks the
// 1-ms code periods in order: every block's length, code ramps and NCO rates depend on the
// previous block's six correlator sums, so a channel is a chain of `ms` dependent steps;
// channels are independent and run side by side on different CUs.
//
// Per block (one loop iteration):
//   map     every lane takes 16 consecutive int8 samples per pass as ONE aligned 16-byte load
//           (a wave reads 1 KiB contiguous), 5 passes of 8 KiB.  The loads of block k+1 are
//           issued before the reduce/filter phases of block k, so HBM latency hides behind them.
//           * code replicas: the three linspace ramps t = fl(fl(i*step)+start) are monotonic
//             and move 0.43 chip over 16 samples, so a group holds at most ONE chip switch
//             (prompt at integer t, early/late together at half-integer t).  Chip index at the
//             group's first sample and the switch sample come from the exact reference
//             arithmetic (an estimate plus two exact probes), which keeps the indices
//             bit-identical to code[int64(ceil(linspace(...)))].
//           * carrier: sample b of a group has phasor G*B_b, G = the lane's group-start phasor
//             (fp64 "turns" reduction + one sincospi per block, then a rotation per pass) and
//             B_b = exp(j b delta) a 16-entry per-block table in LDS.  The lane accumulates
//             sum x_b B_b over the group and over the samples after the switch (5 fp64 ops per
//             sample), then applies G and the code signs once per group.
//           * a group in which early and late switch at different samples (fp64 rounding at a
//             chip boundary) falls back to an exact per-sample loop.
//   reduce  six fp64 partials per lane -> LDS transpose -> 6 waves fold 8 partials each and
//           finish with a DPP wave reduction.
//   filter  wave 0 runs the PLL and prepares the carrier parameters while wave 1 runs the DLL
//           and prepares block size and code ramps, both with the reference's fp64 operation
//           order (-ffp-contract=off; fused multiply-adds only where written as __builtin_fma).
// fp64 everywhere: 1e-7 errors in the sums move the code NCO enough to flip a chip-boundary
// sample somewhere in a 37 s run, which is a 1e-3 relative blip.
#include <math.h>
#include <stdlib.h>

#include "sgx_internal.h"

#define TRK_THREADS 512
#define TRK_PASSES 5                         // ceil((38192+1+15)/16 / 512)
#define TRK_PASS (TRK_THREADS * 16)          // samples per pass (a power of two)

struct TrkConst {
    double fs;
    double code_basis;
    double code_len;
    double spacing;
    double k_code_a;      // tau2code / tau1code
    double k_code_b;      // PDIcode / tau1code
    double k_carr_a;      // tau2carr / tau1carr
    double k_carr_b;      // PDIcarr / tau1carr
    double inv_2pifs_hi;  // 1 / (2 pi fs) as a double-double
    double inv_2pifs_lo;
    double inv_2pi;
    long long rec_len;
    long long rec_alloc;  // bytes that may be read (record + zero pad)
    long long file_off;
    int ms;
    int n_ch;
    int split;            // workgroups cooperating on one channel (1..TRK_PASSES)
    int pad;
};

struct TrkChan {
    double acquiredFreq;
    long long pos0;   // record index of the channel's first sample
    int prn;          // 1-based, 0 = off
    int pad;
};

// Per-block parameters: code part written by wave 1, carrier part by wave 0, read by everybody.
struct TrkBlock {
    long long pos;
    int blk;
    int stop;
    double startE, stepE, startP, stepP, startL, stepL;
    double inv_step;          // ~ 1/step, only used to estimate switch samples
    double r_hi, r_lo;        // carrier turns per sample (double-double)
    double rem_turns;         // carrier phase of sample 0, turns
    double cD, sD;            // rotation over split * TRK_PASS samples (a member's pass stride)
    double2 B[16];            // (cos, sin)(2 pi b r)
};

// Loop state (LDS): code part owned by wave 1, carrier part by wave 0.
struct TrkState {
    double codeFreq, remCode, oldCodeNco, oldCodeErr;
    long long pos;
    double carrFreq, carrBasis, remCarr, w, oldCarrNco, oldCarrErr;
};

__device__ __forceinline__ double ramp_at(int i, double step, double start) {
    return (double)i * step + start;   // two roundings, like numpy's y = arange*step; y += start
}

// chip index at sample ilo and first sample whose chip index is larger (exact reference arithmetic)
__device__ __forceinline__ void ramp_setup(double start, double step, double inv_step, int ilo, int& k1,
                                           int& isw) {
    const double t = ramp_at(ilo, step, start);
    k1 = (int)ceil(t);
    const double kd = (double)k1;
    // t(i) > kd  <=>  i > (kd-start)/step: the estimate is within one sample of the switch
    const int cand = (int)ceil((kd - start) * inv_step);
    const bool at0 = ramp_at(cand, step, start) > kd;
    const bool atm = ramp_at(cand - 1, step, start) > kd;
    isw = at0 ? (atm ? cand - 1 : cand) : cand + 1;
}

// ---- filter phase, code side (wave 1): tracking.py:148-190 scalar part -------------------------------
__device__ __forceinline__ void prep_code(const TrkConst& K, double codeFreq, double rem, long long pos, TrkState& s,
                                          TrkBlock& b, bool writer) {
    const double step = codeFreq / K.fs;                                     // T1
    const int blk = (int)ceil((K.code_len - rem) / step);
    const double nb = (double)blk;
    const double span = nb * step;                                           // blksize * codePhaseStep
    // T3: np.linspace(start, stop, blk, endpoint=False): delta = stop - start; stepL = delta / blk
    const double startE = rem - K.spacing;
    const double stepE = (((span + rem) - K.spacing) - startE) / nb;
    const double startL = rem + K.spacing;
    const double stepL = (((span + rem) + K.spacing) - startL) / nb;
    const double stepP = ((span + rem) - rem) / nb;
    const double t_last = ramp_at(blk - 1, stepP, rem);
    if (writer) {
        b.pos = pos;
        b.blk = blk;
        b.stop = (blk <= 0 || pos + blk > K.rec_len) ? 1 : 0;
        b.startE = startE;
        b.stepE = stepE;
        b.startL = startL;
        b.stepL = stepL;
        b.startP = rem;
        b.stepP = stepP;
        b.inv_step = __builtin_amdgcn_rcp(step);
        s.remCode = (t_last + step) - 1023.0;                                // T4
        s.pos = pos + blk;
    }
}

// ---- filter phase, carrier side (wave 0): parameters of a block with rate w and start phase remCarr -----
__device__ __forceinline__ void prep_carr(const TrkConst& K, double w, double remCarr, TrkBlock& b, int lane) {
    // trigarg = w * (i/fs) + remCarr (T5); in turns: r*i + remCarr/(2 pi), r = w/(2 pi fs)
    const double r_hi = w * K.inv_2pifs_hi;
    const double r_lo = __builtin_fma(w, K.inv_2pifs_hi, -r_hi) + w * K.inv_2pifs_lo;
    // lane b < 16: B_b; lane 16: rotation over a member's pass stride
    const double mult = (lane == 16) ? (double)TRK_PASS * (double)K.split : (double)lane;
    const double p = r_hi * mult;
    const double e = __builtin_fma(r_hi, mult, -p) + r_lo * mult;
    const double u = (p - floor(p)) + e;
    double sn, cs;
    sincospi(2.0 * u, &sn, &cs);
    if (lane < 16) b.B[lane] = make_double2(cs, sn);
    if (lane == 16) {
        b.cD = cs;
        b.sD = sn;
    }
    if (lane == 0) {
        b.r_hi = r_hi;
        b.r_lo = r_lo;
        b.rem_turns = remCarr * K.inv_2pi;
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(ohi, olo);
}

// sum over the 64 lanes of a wave; every lane gets the result (fixed order, deterministic)
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror: every lane of a row of 16 now holds the row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ uint4 load_group(const int8_t* __restrict__ rec, long long addr, long long limit) {
    if (addr > limit) addr = limit;   // never read past the allocation (data of a stopped block is unused)
    return *reinterpret_cast<const uint4*>(rec + addr);
}

__global__ __launch_bounds__(TRK_THREADS) void trk_kernel(const int8_t* __restrict__ rec,
                                                          const int8_t* __restrict__ codes,
                                                          const TrkChan* __restrict__ chans,
                                                          double* __restrict__ out, int* __restrict__ ms_done,
                                                          TrkConst K, long long* __restrict__ prof,
                                                          unsigned long long* __restrict__ xch,
                                                          int* __restrict__ err) {
    __shared__ unsigned s_code_hi[1028];   // hi dword of +-1.0 for [c1022, c0..c1022, c0] (tracking.py:111)
    __shared__ TrkBlock s_blk;
    __shared__ double s_red[6][TRK_THREADS];
    __shared__ double s_tot[6];
    __shared__ TrkState s_st;
    __shared__ unsigned s_gather[TRK_PASSES * 12];

    // optional phase profile (SGX_TRK_PROFILE=1): shader cycles of lane 0 in map / wait / reduce / filter
    long long pf_map = 0, pf_wait = 0, pf_red = 0, pf_flt = 0;

    // block -> (channel, member): members of a channel share blockIdx % 8, i.e. (observed) one XCD / one L2;
    // placement only affects speed, the exchange below is agent-scope and placement independent
    const int P = K.split;
    const int bq = blockIdx.x >> 3, br = blockIdx.x & 7;
    const int ch = br + 8 * (bq / P);
    const int member = bq % P;
    if (ch >= K.n_ch) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const TrkChan cc = chans[ch];
    if (cc.prn == 0) {
        if (tid == 0 && member == 0) ms_done[ch] = 0;
        return;
    }
    unsigned long long* __restrict__ xbase = xch + (long long)ch * 2 * TRK_PASSES * 12;
    for (int i = tid; i < 1028; i += TRK_THREADS) {
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
    if (wave == 0) prep_carr(K, s_st.w, s_st.remCarr, s_blk, lane);
    if (wave == 1) prep_code(K, s_st.codeFreq, s_st.remCode, s_st.pos, s_st, s_blk, lane == 0);
    __syncthreads();

    const long long limit = K.rec_alloc - 16;
    uint4 nx[TRK_PASSES];
    {
        const long long ab = s_blk.pos & ~15ll;
#pragma unroll
        for (int p = 0; p < TRK_PASSES; ++p)
            if (p % P == member) nx[p] = load_group(rec, ab + (long long)(tid + p * TRK_THREADS) * 16, limit);
    }
    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    int done = 0;
    for (int it = 0; it < K.ms; ++it) {
        const long long tk0 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const long long pos = s_blk.pos;
        const int blk = s_blk.blk;
        if (s_blk.stop) break;   // short read: tracking.py:159-163
        const double startE = s_blk.startE, stepE = s_blk.stepE;
        const double startP = s_blk.startP, stepP = s_blk.stepP;
        const double startL = s_blk.startL, stepL = s_blk.stepL;
        const double inv_step = s_blk.inv_step;
        const double cD = s_blk.cD, sD = s_blk.sD;

        const long long abase = pos & ~15ll;
        const int head = (int)(pos - abase);              // bytes of the first group before the block
        const int n_groups = (head + blk + 15) >> 4;
        double aIE = 0.0, aQE = 0.0, aIP = 0.0, aQP = 0.0, aIL = 0.0, aQL = 0.0;
        double gc = 1.0, gs = 0.0;   // carrier phasor at the lane's current group start

#pragma unroll
        for (int p = 0; p < TRK_PASSES; ++p) {
            const int g = tid + p * TRK_THREADS;
            if (p % P == member && g < n_groups) {
                const int i0 = g * 16 - head;             // sample index of byte 0 of this group
                unsigned wd[4] = {nx[p].x, nx[p].y, nx[p].z, nx[p].w};
                if (i0 < 0 || i0 + 16 > blk) {
                    // zero the bytes outside [0, blk): they then add nothing to the sums
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        int lo = -(i0 + 4 * d);
                        lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
                        int hi = i0 + 4 * d + 4 - blk;
                        hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
                        unsigned m = (lo >= 4) ? 0u : (0xFFFFFFFFu << (8 * lo));
                        m &= (hi >= 4) ? 0u : (0xFFFFFFFFu >> (8 * hi));
                        wd[d] &= m;
                    }
                }
                if (p < P) {   // the member's first pass of this block
                    const double r_hi = s_blk.r_hi, r_lo = s_blk.r_lo;
                    const double di0 = (double)i0;
                    const double pr = r_hi * di0;
                    const double er = __builtin_fma(r_hi, di0, -pr);
                    const double u = (pr - floor(pr)) + ((er + r_lo * di0) + s_blk.rem_turns);
                    sincospi(2.0 * u, &gs, &gc);
                } else {
                    const double nc = __builtin_fma(gc, cD, -(gs * sD));
                    const double ns = __builtin_fma(gs, cD, gc * sD);
                    gc = nc;
                    gs = ns;
                }
                const int ilo = i0 < 0 ? 0 : i0;
                int kE, swE, kP, swP, kL, swL;
                ramp_setup(startE, stepE, inv_step, ilo, kE, swE);
                ramp_setup(startP, stepP, inv_step, ilo, kP, swP);
                ramp_setup(startL, stepL, inv_step, ilo, kL, swL);
                const double cE1 = __hiloint2double((int)s_code_hi[kE], 0), cE2 = __hiloint2double((int)s_code_hi[kE + 1], 0);
                const double cP1 = __hiloint2double((int)s_code_hi[kP], 0), cP2 = __hiloint2double((int)s_code_hi[kP + 1], 0);
                const double cL1 = __hiloint2double((int)s_code_hi[kL], 0), cL2 = __hiloint2double((int)s_code_hi[kL + 1], 0);
                const int iend = i0 + 16;
                int swmin = swE < swP ? swE : swP;
                swmin = swL < swmin ? swL : swmin;
                const bool eS = (swE == swmin), pS = (swP == swmin), lS = (swL == swmin);
                const bool odd = (swE < iend && !eS) || (swP < iend && !pS) || (swL < iend && !lS);
                if (__builtin_expect(__any(odd), 0)) {
                    // exact per-sample path (a ramp switches at a second position inside the group)
                    unsigned w0 = wd[0], w1 = wd[1], w2 = wd[2], w3 = wd[3];
#pragma unroll 1
                    for (int b = 0; b < 16; ++b) {
                        const int i = i0 + b;
                        const double xd = (double)(int)(signed char)(w0 & 0xFF);
                        w0 = (w0 >> 8) | (w1 << 24);
                        w1 = (w1 >> 8) | (w2 << 24);
                        w2 = (w2 >> 8) | (w3 << 24);
                        w3 >>= 8;
                        const double2 B = s_blk.B[b];
                        const double c = __builtin_fma(gc, B.x, -(gs * B.y));
                        const double s = __builtin_fma(gs, B.x, gc * B.y);
                        const double xs = s * xd, xc = c * xd;
                        const double cE = i >= swE ? cE2 : cE1;
                        const double cP = i >= swP ? cP2 : cP1;
                        const double cL = i >= swL ? cL2 : cL1;
                        aIE = __builtin_fma(cE, xs, aIE);
                        aQE = __builtin_fma(cE, xc, aQE);
                        aIP = __builtin_fma(cP, xs, aIP);
                        aQP = __builtin_fma(cP, xc, aQP);
                        aIL = __builtin_fma(cL, xs, aIL);
                        aQL = __builtin_fma(cL, xc, aQL);
                    }
                } else {
                    const int bsw = swmin - i0;           // samples b >= bsw come after the switch
                    double Ac = 0.0, As = 0.0, Tc = 0.0, Ts = 0.0;
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        const unsigned wv = wd[b >> 2];
                        const int xi = ((b & 3) == 3) ? ((int)wv >> 24) : (int)(signed char)((wv >> (8 * (b & 3))) & 0xFF);
                        const double xd = (double)xi;
                        const double2 B = s_blk.B[b];
                        Ac = __builtin_fma(xd, B.x, Ac);
                        As = __builtin_fma(xd, B.y, As);
                        const double xt = (b >= bsw) ? xd : 0.0;
                        Tc = __builtin_fma(xt, B.x, Tc);
                        Ts = __builtin_fma(xt, B.y, Ts);
                    }
                    // rotate by the group phasor: cos part -> Q, sin part -> I (tracking.py:205-207)
                    const double allQ = __builtin_fma(gc, Ac, -(gs * As));
                    const double allI = __builtin_fma(gs, Ac, gc * As);
                    const double tlQ = __builtin_fma(gc, Tc, -(gs * Ts));
                    const double tlI = __builtin_fma(gs, Tc, gc * Ts);
                    const double dE = eS ? (cE2 - cE1) : 0.0;
                    const double dP = pS ? (cP2 - cP1) : 0.0;
                    const double dL = lS ? (cL2 - cL1) : 0.0;
                    aIE = __builtin_fma(dE, tlI, __builtin_fma(cE1, allI, aIE));
                    aQE = __builtin_fma(dE, tlQ, __builtin_fma(cE1, allQ, aQE));
                    aIP = __builtin_fma(dP, tlI, __builtin_fma(cP1, allI, aIP));
                    aQP = __builtin_fma(dP, tlQ, __builtin_fma(cP1, allQ, aQP));
                    aIL = __builtin_fma(dL, tlI, __builtin_fma(cL1, allI, aIL));
                    aQL = __builtin_fma(dL, tlQ, __builtin_fma(cL1, allQ, aQL));
                }
            }
        }
        // prefetch the next block's groups: they land during the reduce and filter phases
        {
            const long long ab = (pos + blk) & ~15ll;
#pragma unroll
            for (int p = 0; p < TRK_PASSES; ++p)
                if (p % P == member) nx[p] = load_group(rec, ab + (long long)(tid + p * TRK_THREADS) * 16, limit);
        }
        const long long tk1 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        s_red[0][tid] = aIE;
        s_red[1][tid] = aQE;
        s_red[2][tid] = aIP;
        s_red[3][tid] = aQP;
        s_red[4][tid] = aIL;
        s_red[5][tid] = aQL;
        __syncthreads();
        const long long tk2 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (wave < 6) {
            double acc = s_red[wave][lane];
#pragma unroll
            for (int k = 1; k < TRK_THREADS / 64; ++k) acc += s_red[wave][lane + 64 * k];
            acc = wave_sum(acc);
            if (P == 1) {
                if (lane == 0) s_tot[wave] = acc;
            } else if (lane == 0) {
                // publish this member's partial as two {epoch, 32-bit payload} granules: ONE aligned 8-byte
                // write-through (sc1) store each, so a reader never sees a torn granule (Guideline 16, R2)
                const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
                unsigned long long* gp = xbase + ((it & 1) * TRK_PASSES + member) * 12 + 2 * wave;
                __hip_atomic_store(gp, tag | (unsigned)__double2loint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gp + 1, tag | (unsigned)__double2hiint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (P > 1 && wave == 0) {
                // gather every member's granules of this epoch (relaxed agent-scope polls, bounded)
                const unsigned epoch = (unsigned)(it + 1);
                const unsigned long long* gp = xbase + (it & 1) * TRK_PASSES * 12;
                const bool mine = lane < 12 * P;
                unsigned long long x = 0;
                int budget = 1 << 22;
                for (;;) {
                    if (mine) x = __hip_atomic_load(gp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ok = !mine || (unsigned)(x >> 32) == epoch;
                    if (__all(ok)) break;
                    if (--budget == 0) {
                        if (lane == 0) atomicExch(err, 1 + ch);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (mine) s_gather[lane] = (unsigned)x;
                // same wave: LDS accesses are in order
                if (lane < 6) {
                    double t = 0.0;
                    for (int c = 0; c < P; ++c)
                        t += __hiloint2double((int)s_gather[c * 12 + 2 * lane + 1], (int)s_gather[c * 12 + 2 * lane]);
                    s_tot[lane] = t;
                }
            }
        }
        __syncthreads();
        const long long tk3 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const long long m = K.ms;
        const bool more = (it + 1 < K.ms);
        if (wave == 0) {
            // T7 PLL (tracking.py:223-235) and carrier bookkeeping (T5) for the next block
            const double I_P = s_tot[2], Q_P = s_tot[3];
            const double w_old = s_st.w, rem_old = s_st.remCarr;
            const double oldNco = s_st.oldCarrNco, oldErr = s_st.oldCarrErr, basis = s_st.carrBasis;
            // remCarrPhase = trigarg[blk] % (2 pi) with trigarg = w*(blk/fs) + rem  (exact remainder by FMA)
            const double two_pi = 2 * M_PI;
            const double arg_end = w_old * ((double)blk / K.fs) + rem_old;
            double kq = floor(arg_end / two_pi);
            double rc = __builtin_fma(-kq, two_pi, arg_end);
            if (rc < 0.0) rc += two_pi;
            if (rc >= two_pi) rc -= two_pi;
            const double carrError = atan(Q_P / I_P) / 2.0 / M_PI;
            const double carrNco = oldNco + K.k_carr_a * (carrError - oldErr) + carrError * K.k_carr_b;
            const double carrFreq = basis + carrNco;
            const double w_new = (carrFreq * 2.0) * M_PI;
            if (more) prep_carr(K, w_new, rc, s_blk, lane);
            const bool rec_out = (member == 0);
            if (lane == 0) {
                s_st.w = w_new;
                s_st.remCarr = rc;
                s_st.oldCarrNco = carrNco;
                s_st.oldCarrErr = carrError;
                s_st.carrFreq = carrFreq;
            }
            if (lane == 0 && rec_out) {
                o[2 * m + it] = carrFreq;          // T9 record (tracking.py:255-275)
                o[3 * m + it] = I_P;
                o[4 * m + it] = s_tot[0];
                o[5 * m + it] = s_tot[4];
                o[6 * m + it] = s_tot[1];
                o[7 * m + it] = Q_P;
                o[8 * m + it] = s_tot[5];
                o[11 * m + it] = carrError;
                o[12 * m + it] = carrNco;
            }
        } else if (wave == 1) {
            // T8 DLL (tracking.py:238-251), then block size and ramps of the next block (T1, T3, T4)
            const double I_E = s_tot[0], Q_E = s_tot[1], I_L = s_tot[4], Q_L = s_tot[5];
            const double oldNco = s_st.oldCodeNco, oldErr = s_st.oldCodeErr;
            const long long pos_after = s_st.pos;
            const double rem_next = s_st.remCode;
            const double eE = sqrt(I_E * I_E + Q_E * Q_E);
            const double eL = sqrt(I_L * I_L + Q_L * Q_L);
            const double codeError = (eE - eL) / (eE + eL);
            const double codeNco = oldNco + K.k_code_a * (codeError - oldErr) + codeError * K.k_code_b;
            const double codeFreq = K.code_basis - codeNco;
            if (lane == 0) {
                s_st.oldCodeNco = codeNco;
                s_st.oldCodeErr = codeError;
                s_st.codeFreq = codeFreq;
            }
            if (lane == 0 && member == 0) {
                o[0 * m + it] = (double)(pos_after + K.file_off);
                o[1 * m + it] = codeFreq;
                o[9 * m + it] = codeError;
                o[10 * m + it] = codeNco;
            }
            if (more) prep_code(K, codeFreq, rem_next, pos_after, s_st, s_blk, lane == 0);
        }
        done = it + 1;
        __syncthreads();   // next block's parameters visible
        if (prof && tid == 0) {
            const long long tk4 = (long long)__builtin_amdgcn_s_memtime();
            pf_map += tk1 - tk0;
            pf_wait += tk2 - tk1;
            pf_red += tk3 - tk2;
            pf_flt += tk4 - tk3;
        }
    }
    if (tid == 0 && prof && member == 0) {
        prof[ch * 4 + 0] = pf_map;
        prof[ch * 4 + 1] = pf_wait;
        prof[ch * 4 + 2] = pf_red;
        prof[ch * 4 + 3] = pf_flt;
    }
    if (tid == 0 && member == 0) ms_done[ch] = done;
}

// tracking.py:65-94: series start as zeros (absoluteSample, I/Q) or +Inf (the others)
__global__ __launch_bounds__(256) void trk_fill_kernel(double* __restrict__ out, long long ms, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int series = (int)((i / ms) % SGX_NUM_SERIES);
    const bool zero = (series == 0) || (series >= 3 && series <= 8);
    out[i] = zero ? 0.0 : __longlong_as_double(0x7FF0000000000000ll);
}

extern "C" int sgx_track(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch,
                         int32_t n_ch, int32_t ms, double* out, int32_t* ms_done) {
    SGX_CHECK_ARG(c && r && ch && out && ms_done);
    SGX_CHECK_ARG(n_ch >= 1 && n_ch <= 65535 && ms >= 1);
    SGX_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const sgx_settings& S = c->s;
    if (c->n_code + 1 + 15 > (long long)TRK_PASSES * TRK_PASS) {
        sgx_set_error("samplesPerCode %lld exceeds the tracking kernel's %d samples per block", (long long)c->n_code,
                      TRK_PASSES * TRK_PASS - 16);
        return SGX_E_ARG;
    }

    TrkConst K;
    K.fs = S.samplingFreq;
    K.code_basis = S.codeFreqBasis;
    K.code_len = (double)S.codeLength;
    K.spacing = S.dllCorrelatorSpacing;
    double t1c, t2c, t1p, t2p;
    sgx_calc_loop_coef(S.dllNoiseBandwidth, S.dllDampingRatio, 1.0, &t1c, &t2c);     // tracking.py:45
    sgx_calc_loop_coef(S.pllNoiseBandwidth, S.pllDampingRatio, 0.25, &t1p, &t2p);    // tracking.py:52
    K.k_code_a = t2c / t1c;
    K.k_code_b = 0.001 / t1c;
    K.k_carr_a = t2p / t1p;
    K.k_carr_b = 0.001 / t1p;
    {
        const long double two_pi = 2.0L * (long double)M_PI;   // the reference's 2*np.pi (a double)
        const long double inv = 1.0L / (two_pi * (long double)S.samplingFreq);
        K.inv_2pifs_hi = (double)inv;
        K.inv_2pifs_lo = (double)(inv - (long double)K.inv_2pifs_hi);
        K.inv_2pi = (double)(1.0L / two_pi);
    }
    K.rec_len = (long long)r->n;
    K.rec_alloc = (long long)r->n + SGX_IF_PAD;
    K.file_off = rec_file_offset;
    K.ms = ms;
    K.n_ch = n_ch;
    K.pad = 0;
    {
        // cooperating workgroups per channel: one 512-thread workgroup fills a CU, all of a launch must be
        // resident at once (they wait for each other), so split * n_ch <= CU count
        int cus = 0;
        SGX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        int split = cus / (n_ch > 0 ? ((n_ch + 7) / 8) * 8 : 8);
        if (split > TRK_PASSES) split = TRK_PASSES;
        if (split < 1) split = 1;
        const char* se = getenv("SGX_TRK_SPLIT");
        if (se && atoi(se) >= 1 && atoi(se) <= split) split = atoi(se);
        K.split = split;
    }

    std::vector<TrkChan> hc((size_t)n_ch);
    for (int i = 0; i < n_ch; ++i) {
        hc[(size_t)i].acquiredFreq = ch[i].acquiredFreq;
        hc[(size_t)i].prn = ch[i].prn;
        hc[(size_t)i].pad = 0;
        SGX_CHECK_ARG(ch[i].prn >= 0 && ch[i].prn <= 32);
        const long long p0 = (long long)S.skipNumberOfBytes + (long long)ch[i].codePhase - rec_file_offset;
        if (ch[i].prn != 0 && p0 < 0) {
            sgx_set_error("channel %d starts at file byte %lld, before the record (offset %lld)", i,
                          (long long)S.skipNumberOfBytes + (long long)ch[i].codePhase, (long long)rec_file_offset);
            return SGX_E_RANGE;
        }
        hc[(size_t)i].pos0 = p0;
    }
    const size_t elems = (size_t)n_ch * SGX_NUM_SERIES * (size_t)ms;
    if (c->trk_out_elems < elems) {
        if (c->d_trk_out) hipFree(c->d_trk_out);
        c->d_trk_out = nullptr;
        c->trk_out_elems = 0;
        hipError_t e = hipMalloc((void**)&c->d_trk_out, elems * sizeof(double));
        if (e != hipSuccess) {
            sgx_set_error("hipMalloc of %zu tracking output bytes failed", elems * sizeof(double));
            return SGX_E_NOMEM;
        }
        c->trk_out_elems = elems;
    }
    TrkChan* d_ch = nullptr;
    int* d_done = nullptr;
    SGX_HIP(hipMalloc((void**)&d_ch, sizeof(TrkChan) * (size_t)n_ch));
    SGX_HIP(hipMalloc((void**)&d_done, sizeof(int) * (size_t)n_ch));
    SGX_HIP(hipMemcpyAsync(d_ch, hc.data(), sizeof(TrkChan) * (size_t)n_ch, hipMemcpyHostToDevice, st));
    SGX_HIP(hipMemsetAsync(d_done, 0, sizeof(int) * (size_t)n_ch, st));
    trk_fill_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, st>>>(c->d_trk_out, ms, (long long)elems);
    long long* d_prof = nullptr;
    const char* pe = getenv("SGX_TRK_PROFILE");
    const bool want_prof = pe && pe[0] == '1';
    if (want_prof) SGX_HIP(hipMalloc((void**)&d_prof, sizeof(long long) * 4 * (size_t)n_ch));
    unsigned long long* d_xch = nullptr;
    int* d_err = nullptr;
    const size_t xch_bytes = sizeof(unsigned long long) * (size_t)n_ch * 2 * TRK_PASSES * 12;
    SGX_HIP(hipMalloc((void**)&d_xch, xch_bytes));
    SGX_HIP(hipMalloc((void**)&d_err, sizeof(int)));
    SGX_HIP(hipMemsetAsync(d_xch, 0, xch_bytes, st));   // every polled word is zeroed before every launch
    SGX_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    const int n_blocks = ((n_ch + 7) / 8) * 8 * K.split;
    hipEventRecord(c->ev[3], st);
    trk_kernel<<<n_blocks, TRK_THREADS, 0, st>>>(r->d, c->d_codes, d_ch, c->d_trk_out, d_done, K, d_prof, d_xch,
                                                 d_err);
    hipEventRecord(c->ev[4], st);
    hipError_t e = hipGetLastError();
    int h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out, c->d_trk_out, elems * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ms_done, d_done, sizeof(int) * (size_t)n_ch, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (want_prof && e == hipSuccess) {
        std::vector<long long> hp(4 * (size_t)n_ch);
        hipMemcpy(hp.data(), d_prof, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost);
        for (int i = 0; i < n_ch && i < 4; ++i)
            fprintf(stderr, "[sgx trk profile] ch %d cycles/block: map %.0f wait %.0f reduce %.0f filter %.0f\n", i,
                    (double)hp[4 * i] / ms, (double)hp[4 * i + 1] / ms, (double)hp[4 * i + 2] / ms,
                    (double)hp[4 * i + 3] / ms);
    }
    if (d_prof) hipFree(d_prof);
    hipFree(d_ch);
    hipFree(d_done);
    hipFree(d_xch);
    hipFree(d_err);
    if (e != hipSuccess) {
        sgx_set_error("tracking kernel failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    if (h_err != 0) {
        sgx_set_error("tracking kernel: channel %d timed out waiting for a cooperating workgroup (split %d); "
                      "set SGX_TRK_SPLIT=1", h_err - 1, K.split);
        return SGX_E_HIP;
    }
    hipEventElapsedTime(&c->timing.track_ms, c->ev[3], c->ev[4]);
    return SGX_OK;
}
