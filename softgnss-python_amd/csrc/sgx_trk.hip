// TrackingResult.track on gfx950 (reference tracking.py:13-295).
//
// One persistent 512-thread workgroup (8 waves, two per SIMD of one CU) per channel walks the 1-ms code
// periods in order: every block's length, code ramps and NCO rates depend on the previous
// block's six correlator sums, so a channel is a chain of `ms` dependent steps; channels are
// independent and run side by side on different CUs.
//
// Per block:
//   * lane 0 (the "loop filter" lane) turns the six sums into the next block's parameters with
//     the reference's exact fp64 operation order (this file is built with -ffp-contract=off;
//     fused multiply-adds appear only where written as __builtin_fma);
//   * every lane takes 16 consecutive int8 samples per pass as ONE aligned 16-byte load (wave =
//     1 KiB contiguous, fully coalesced), 512 lanes * 16 B = 8 KiB per pass, 5 passes (93 % of
//     the lane slots carry samples; 1024 lanes would cap at 128 VGPRs and spill);
//   * code replicas: the three linspace ramps t = fl(fl(i*step)+start) are monotonic and move
//     0.43 chip over 16 samples, so each ramp switches chip at most once inside a group.  The
//     chip index at the group's first sample and the switch sample are found with the exact
//     reference arithmetic (an estimate plus three exact probes); the 16 samples then only pick
//     between two code values.  This keeps chip indices bit-identical to
//     code[int64(ceil(linspace(...)))] at a fraction of the per-sample cost;
//   * carrier: the lane's start phase is reduced in fp64 "turns" (double-double rate), one
//     sincospi, then a complex rotation per sample (4 FMAs) instead of a large-argument sin/cos;
//   * six fp64 accumulators per lane -> LDS transpose -> 6 waves fold 16 partials and finish with
//     a wave butterfly.  fp64 everywhere: 1e-7 errors in the sums move the code NCO enough to
//     flip a chip-boundary sample somewhere in a 37 s run, which is a 1e-3 relative blip.
#include <math.h>

#include "sgx_internal.h"

#define TRK_THREADS 512
#define TRK_GROUP 16                         // samples per lane per pass (one dwordx4)
#define TRK_PASS (TRK_THREADS * TRK_GROUP)   // samples per pass

struct TrkConst {
    double fs;
    double code_basis;
    double code_len;
    double spacing;
    double k_code_a;   // tau2code / tau1code
    double k_code_b;   // PDIcode / tau1code
    double k_carr_a;   // tau2carr / tau1carr
    double k_carr_b;   // PDIcarr / tau1carr
    long long rec_len;
    long long file_off;
    int ms;
    int n_ch;
};

struct TrkChan {
    double acquiredFreq;
    long long pos0;   // record index of the channel's first sample
    int prn;          // 1-based, 0 = off
    int pad;
};

// Per-block parameters, written by lane 0, read by everybody.
struct TrkBlock {
    long long pos;
    int blk;
    int stop;
    double startE, stepE, startP, stepP, startL, stepL;
    double inv_step;          // ~ 1/stepP, only used to estimate switch samples
    double r_hi, r_lo;        // carrier turns per sample (double-double)
    double rem_turns;         // carrier phase of sample 0, turns
    double cd, sd;            // one-sample rotation
    double cD, sD;            // TRK_PASS-sample rotation
};

// Loop state owned by lane 0.
struct TrkState {
    double codeFreq, remCode, carrFreq, carrBasis, remCarr;
    double oldCodeNco, oldCodeErr, oldCarrNco, oldCarrErr;
    long long pos;
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
    const int cand = (int)ceil((kd - start) * inv_step);
    isw = cand + 2;
    if (ramp_at(cand + 1, step, start) > kd) isw = cand + 1;
    if (ramp_at(cand, step, start) > kd) isw = cand;
    if (ramp_at(cand - 1, step, start) > kd) isw = cand - 1;
}

// tracking.py:148-201 scalar part: block size, the three linspace ramps, carrier phase bookkeeping
__device__ void trk_prepare(const TrkConst& K, TrkState& s, TrkBlock& b) {
    const double step = s.codeFreq / K.fs;                                   // T1
    const int blk = (int)ceil((K.code_len - s.remCode) / step);
    b.pos = s.pos;
    b.blk = blk;
    b.stop = (blk <= 0 || s.pos + blk > K.rec_len) ? 1 : 0;
    const double nb = (double)blk;
    const double span = nb * step;                                           // blksize * codePhaseStep
    // T3: np.linspace(start, stop, blk, endpoint=False): delta = stop - start; stepL = delta / blk
    b.startE = s.remCode - K.spacing;
    b.stepE = (((span + s.remCode) - K.spacing) - b.startE) / nb;
    b.startL = s.remCode + K.spacing;
    b.stepL = (((span + s.remCode) + K.spacing) - b.startL) / nb;
    b.startP = s.remCode;
    b.stepP = ((span + s.remCode) - b.startP) / nb;
    b.inv_step = 1.0 / step;
    // T5: trigarg = ((carrFreq*2.0)*pi) * (i/fs) + remCarrPhase
    const double w = (s.carrFreq * 2.0) * M_PI;
    const double two_pi = 2 * M_PI;
    const double a = w / two_pi;                                             // cycles per second
    b.r_hi = a / K.fs;
    b.r_lo = __builtin_fma(-b.r_hi, K.fs, a) / K.fs;
    b.rem_turns = s.remCarr / two_pi;
    sincospi(2.0 * b.r_hi, &b.sd, &b.cd);
    const double big = b.r_hi * (double)TRK_PASS;                            // exact (power of two)
    const double frac = (big - floor(big)) + b.r_lo * (double)TRK_PASS;
    sincospi(2.0 * frac, &b.sD, &b.cD);
    // state that does not need the sums: T4 remCodePhase, T5 remCarrPhase, file position
    const double t_last = ramp_at(blk - 1, b.stepP, b.startP);
    s.remCode = (t_last + step) - 1023.0;
    const double arg_end = w * (nb / K.fs) + s.remCarr;
    double rc = fmod(arg_end, two_pi);
    if (rc < 0.0) rc += two_pi;
    s.remCarr = rc;
    s.pos += blk;
}

__global__ __launch_bounds__(TRK_THREADS) void trk_kernel(const int8_t* __restrict__ rec,
                                                          const int8_t* __restrict__ codes,
                                                          const TrkChan* __restrict__ chans,
                                                          double* __restrict__ out, int* __restrict__ ms_done,
                                                          TrkConst K) {
    __shared__ unsigned s_code_hi[1028];   // hi dword of +-1.0 for [c1022, c0..c1022, c0] (tracking.py:111)
    __shared__ TrkBlock s_blk;
    __shared__ double s_red[6][TRK_THREADS];
    __shared__ double s_tot[6];
    __shared__ TrkState s_st;              // loop state, touched by lane 0 only

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    const TrkChan cc = chans[ch];
    if (cc.prn == 0) {
        if (tid == 0) ms_done[ch] = 0;
        return;
    }
    for (int i = tid; i < 1028; i += TRK_THREADS) {
        int j = i - 1;
        if (j < 0) j = 1022;
        if (j >= 1023) j -= 1023;
        if (j >= 1023) j -= 1023;
        s_code_hi[i] = (codes[(cc.prn - 1) * 1023 + j] > 0) ? 0x3FF00000u : 0xBFF00000u;
    }
    TrkState& st = s_st;
    if (tid == 0) {
        st.codeFreq = K.code_basis;   // tracking.py:114-130
        st.remCode = 0.0;
        st.carrFreq = cc.acquiredFreq;
        st.carrBasis = cc.acquiredFreq;
        st.remCarr = 0.0;
        st.oldCodeNco = st.oldCodeErr = st.oldCarrNco = st.oldCarrErr = 0.0;
        st.pos = cc.pos0;
        trk_prepare(K, st, s_blk);
    }
    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    int done = 0;
    for (int it = 0; it < K.ms; ++it) {
        __syncthreads();   // s_blk (and on the first pass s_code_hi) visible
        const long long pos = s_blk.pos;
        const int blk = s_blk.blk;
        if (s_blk.stop) break;   // short read: tracking.py:159-163
        const double startE = s_blk.startE, stepE = s_blk.stepE;
        const double startP = s_blk.startP, stepP = s_blk.stepP;
        const double startL = s_blk.startL, stepL = s_blk.stepL;
        const double inv_step = s_blk.inv_step;
        const double r_hi = s_blk.r_hi, r_lo = s_blk.r_lo, rem_turns = s_blk.rem_turns;
        const double cd = s_blk.cd, sd = s_blk.sd, cD = s_blk.cD, sD = s_blk.sD;

        const long long abase = pos & ~15ll;
        const int head = (int)(pos - abase);              // samples of the first group before the block
        const int n_groups = (head + blk + 15) >> 4;
        double aIE = 0.0, aQE = 0.0, aIP = 0.0, aQP = 0.0, aIL = 0.0, aQL = 0.0;
        double gc = 1.0, gs = 0.0;   // carrier phasor at the lane's current group start

        for (int p = 0, g = tid; g < n_groups; ++p, g += TRK_THREADS) {
            const int i0 = g * 16 - head;                 // sample index of byte 0 of this group
            const uint4 w4 = *reinterpret_cast<const uint4*>(rec + abase + (long long)g * 16);
            unsigned wd[4] = {w4.x, w4.y, w4.z, w4.w};
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
            if (p == 0) {
                const double di0 = (double)i0;
                const double pr = r_hi * di0;
                const double er = __builtin_fma(r_hi, di0, -pr);
                const double u = (pr - floor(pr)) + ((er + r_lo * di0) + rem_turns);
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
            const unsigned hE1 = s_code_hi[kE], hE2 = s_code_hi[kE + 1];
            const unsigned hP1 = s_code_hi[kP], hP2 = s_code_hi[kP + 1];
            const unsigned hL1 = s_code_hi[kL], hL2 = s_code_hi[kL + 1];
            double c = gc, s = gs;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int i = i0 + b;
                const int xi = (int)(signed char)((wd[b >> 2] >> (8 * (b & 3))) & 0xFF);
                const double xd = (double)xi;
                const double xs = s * xd;   // iBasebandSignal = carrSin * raw (tracking.py:207)
                const double xc = c * xd;   // qBasebandSignal = carrCos * raw (tracking.py:205)
                const double cE = __hiloint2double((int)(i >= swE ? hE2 : hE1), 0);
                const double cP = __hiloint2double((int)(i >= swP ? hP2 : hP1), 0);
                const double cL = __hiloint2double((int)(i >= swL ? hL2 : hL1), 0);
                aIE = __builtin_fma(cE, xs, aIE);
                aQE = __builtin_fma(cE, xc, aQE);
                aIP = __builtin_fma(cP, xs, aIP);
                aQP = __builtin_fma(cP, xc, aQP);
                aIL = __builtin_fma(cL, xs, aIL);
                aQL = __builtin_fma(cL, xc, aQL);
                const double nc = __builtin_fma(c, cd, -(s * sd));
                const double ns = __builtin_fma(s, cd, c * sd);
                c = nc;
                s = ns;
            }
        }
        s_red[0][tid] = aIE;
        s_red[1][tid] = aQE;
        s_red[2][tid] = aIP;
        s_red[3][tid] = aQP;
        s_red[4][tid] = aIL;
        s_red[5][tid] = aQL;
        __syncthreads();
        if (tid < 6 * 64) {
            const int v = tid >> 6, l = tid & 63;
            double acc = s_red[v][l];
#pragma unroll
            for (int k = 1; k < TRK_THREADS / 64; ++k) acc += s_red[v][l + 64 * k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (l == 0) s_tot[v] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            const double I_E = s_tot[0], Q_E = s_tot[1], I_P = s_tot[2], Q_P = s_tot[3], I_L = s_tot[4],
                         Q_L = s_tot[5];
            // T7 PLL (tracking.py:223-235)
            const double carrError = atan(Q_P / I_P) / 2.0 / M_PI;
            const double carrNco = st.oldCarrNco + K.k_carr_a * (carrError - st.oldCarrErr) + carrError * K.k_carr_b;
            st.oldCarrNco = carrNco;
            st.oldCarrErr = carrError;
            st.carrFreq = st.carrBasis + carrNco;
            // T8 DLL (tracking.py:238-251)
            const double eE = sqrt(I_E * I_E + Q_E * Q_E);
            const double eL = sqrt(I_L * I_L + Q_L * Q_L);
            const double codeError = (eE - eL) / (eE + eL);
            const double codeNco = st.oldCodeNco + K.k_code_a * (codeError - st.oldCodeErr) + codeError * K.k_code_b;
            st.oldCodeNco = codeNco;
            st.oldCodeErr = codeError;
            st.codeFreq = K.code_basis - codeNco;
            // T9 record (tracking.py:255-275); st.pos already points past this block
            const long long m = K.ms;
            o[0 * m + it] = (double)(st.pos + K.file_off);
            o[1 * m + it] = st.codeFreq;
            o[2 * m + it] = st.carrFreq;
            o[3 * m + it] = I_P;
            o[4 * m + it] = I_E;
            o[5 * m + it] = I_L;
            o[6 * m + it] = Q_E;
            o[7 * m + it] = Q_P;
            o[8 * m + it] = Q_L;
            o[9 * m + it] = codeError;
            o[10 * m + it] = codeNco;
            o[11 * m + it] = carrError;
            o[12 * m + it] = carrNco;
            done = it + 1;
            if (it + 1 < K.ms) trk_prepare(K, st, s_blk);
        }
    }
    if (tid == 0) ms_done[ch] = done;
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
    K.rec_len = (long long)r->n;
    K.file_off = rec_file_offset;
    K.ms = ms;
    K.n_ch = n_ch;

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
    hipEventRecord(c->ev[3], st);
    trk_kernel<<<n_ch, TRK_THREADS, 0, st>>>(r->d, c->d_codes, d_ch, c->d_trk_out, d_done, K);
    hipEventRecord(c->ev[4], st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, c->d_trk_out, elems * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ms_done, d_done, sizeof(int) * (size_t)n_ch, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_ch);
    hipFree(d_done);
    if (e != hipSuccess) {
        sgx_set_error("tracking kernel failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    hipEventElapsedTime(&c->timing.track_ms, c->ev[3], c->ev[4]);
    return SGX_OK;
}
