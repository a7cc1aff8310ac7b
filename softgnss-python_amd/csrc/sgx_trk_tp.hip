// Throughput-mode tracking kernel for gfx950: one 256-thread workgroup per channel, two workgroups per CU,
// used when there are more channels than CUs (split == 1).  Same arithmetic contract as sgx_trk.hip
// (reference tracking.py:13-295) but a different map, chosen for instruction count
// rather than latency - in this regime the chip is bound by fp64 VALU issue, not by the dependency chain:
//
//   ONE LANE PER PROMPT CHIP.  Lane c takes the samples whose prompt index ceil(tP) equals c: [s0, s1), both
//   ends found with the exact reference arithmetic (estimate from 1/step plus two exact probes).  The prompt
//   code is constant there, and the early and late ramps (spanning < 1 chip) switch at most once each, at
//   eE and eL (normally the same sample, mid-chip).  So the lane's ~37 samples are a HEAD run [s0, e1) and a
//   TAIL run [e2, s1) with all three codes constant in each: per sample only int8 -> fp64 and two FMAs into
//   the run's accumulator (phasor table B_k in registers), no per-sample selects.  Bytes are fetched at
//   dword alignment and realigned with v_alignbyte; bytes past a run's end are masked once per run.
//   Runs are rotated by the run-start phasor (four small tables + one table lookup) and the code signs are
//   applied once per lane.  A chip whose early and late switches differ (fp64 rounding exactly at a boundary)
//   or whose runs exceed 20 samples takes an exact per-sample loop.
//   ~320 instructions per 37 samples instead of ~300 per 16 in the group kernel.
#include "sgx_trk_common.h"

// first sample above thr from real arithmetic; `near` is raised when a sample lies within 1e-7 samples of the boundary
__device__ __forceinline__ int tp_bound(double start, double inv_step, double thr, bool& near) {
    const double u = (thr - start) * inv_step;
    const double f = floor(u);
    const double fr = u - f;
    near = near || !(fr > 1e-7 && fr < 1.0 - 1e-7);
    return (int)f + 1;
}

// 20 bytes starting at record byte `addr` (any alignment; the hardware takes unaligned 16-byte loads), bytes >= len
// zeroed: five dwords
__device__ __forceinline__ void load_run_u(const int8_t* __restrict__ rec, long long addr, long long limit, int len,
                                           unsigned (&w)[5]) {
    if (addr > limit) addr = limit;
    const U4a q = *reinterpret_cast<const U4a*>(rec + addr);
    const unsigned q4 = reinterpret_cast<const U2a*>(rec + addr + 16)->x;
    w[0] = q.x;
    w[1] = q.y;
    w[2] = q.z;
    w[3] = q.w;
    w[4] = q4;
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        int keep = len - 4 * d;                      // bytes of this dword inside the run
        keep = keep < 0 ? 0 : (keep > 4 ? 4 : keep);
        w[d] &= (keep >= 4) ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u);
    }
}

__global__ __launch_bounds__(TRK_THREADS, 3) void trk_kernel_tp(const int8_t* __restrict__ rec,
                                                                const int8_t* __restrict__ codes,
                                                                const TrkChan* __restrict__ chans,
                                                                double* __restrict__ out, int* __restrict__ ms_done,
                                                                TrkConst K) {
    __shared__ unsigned s_code_hi[1028];   // hi dword of +-1.0 for [c1022, c0..c1022, c0] (tracking.py:111)
    __shared__ TrkBlock s_blk;             // code part used; carrier part unused here
    __shared__ TpCarr s_car;
    __shared__ double s_red[6][TRK_THREADS];
    __shared__ double s_tot[6];
    __shared__ TrkState s_st;

    const int ch = blockIdx.x;
    if (ch >= K.n_ch) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
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
    }
    if (wave == 1) prep_code(K, s_st.codeFreq, s_st.remCode, s_st.pos, s_st, s_blk, lane == 0);
    __syncthreads();

    const long long limit = K.rec_alloc - 24;
    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    const long long m = K.ms;
    const double two_pi = 2 * M_PI;
    int done = 0;
    for (int it = 0; it < K.ms; ++it) {
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
#pragma unroll 1
        for (int c = c_first + tid; c <= c_last; c += TRK_THREADS) {
            asm volatile("" ::: "memory");   // the slot phasors are re-read from LDS for every chip (80 registers otherwise)
            // ---- the chip's sample range and the early / late switch samples (exact) ----
            // boundaries from one multiply each, u = (thr - start) / step in real arithmetic: the reference's ramp
            // fl(fl(i step) + start) lies within 1e-11 samples of the real one, so floor(u) + 1 is the first sample above
            // thr unless u is within 1e-7 of an integer - then (any lane of the wave) the exact probes decide
            bool near = false;
            int s0 = (c == c_first) ? 0 : tp_bound(startP, inv_step, (double)(c - 1), near);
            int s1 = (c == c_last) ? blk : tp_bound(startP, inv_step, (double)c, near);
            if (__builtin_expect(__any(near), 0)) {
                s0 = (c == c_first) ? 0 : first_above(startP, stepP, inv_step, (double)(c - 1));
                s1 = (c == c_last) ? blk : first_above(startP, stepP, inv_step, (double)c);
            }
            s0 = s0 < 0 ? 0 : s0;
            s1 = s1 > blk ? blk : s1;
            const int kE = (int)ceil(ramp_at(s0, stepE, startE));
            const int kL = (int)ceil(ramp_at(s0, stepL, startL));
            near = false;
            int eE = tp_bound(startE, inv_step, (double)kE, near);
            int eL = tp_bound(startL, inv_step, (double)kL, near);
            if (__builtin_expect(__any(near), 0)) {
                eE = first_above(startE, stepE, inv_step, (double)kE);
                eL = first_above(startL, stepL, inv_step, (double)kL);
            }
            eE = eE > s1 ? s1 : eE;
            eL = eL > s1 ? s1 : eL;
            const int e1 = eE < eL ? eE : eL, e2 = eE < eL ? eL : eE;
            const int len_h = e1 - s0, len_t = s1 - e2;
            const double cP = __hiloint2double((int)s_code_hi[c], 0);
            const double cEh = __hiloint2double((int)s_code_hi[kE], 0), cEn = __hiloint2double((int)s_code_hi[kE + 1], 0);
            const double cLh = __hiloint2double((int)s_code_hi[kL], 0), cLn = __hiloint2double((int)s_code_hi[kL + 1], 0);
            // run-start phasor of the head from the four tables
            const double2 gh = cmul2(cmul2(s_car.W3[s0 >> 12], s_car.W2[(s0 >> 8) & 15]),
                                     cmul2(s_car.W1[(s0 >> 4) & 15], s_car.B[s0 & 15]));
            const bool odd = (s1 > s0) && (e2 != e1 || len_h > TP_RUN || len_t > TP_RUN || e2 - s0 > 31);
            if (__builtin_expect(__any(odd), 0)) {
                // exact per-sample loop over the chip (rare)
                if (s1 > s0) {
                    double2 ph = gh;
                    const double2 b1 = s_car.B[1];
                    for (int i = s0; i < s1; ++i) {
                        long long a = pos + i;
                        const double xd = (double)(int)rec[a > K.rec_alloc - 1 ? K.rec_alloc - 1 : a];
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
            } else if (s1 > s0) {
                unsigned wh[5], wt[5];
                load_run_u(rec, pos + s0, limit, len_h, wh);
                load_run_u(rec, pos + e2, limit, len_t, wt);
                const double2 gt = cmul2(gh, s_car.B[e2 - s0]);
                double Hc = 0.0, Hs = 0.0, Tc = 0.0, Ts = 0.0;
#pragma unroll
                for (int k = 0; k < TP_RUN; ++k) {
                    const unsigned a = wh[k >> 2], b = wt[k >> 2];
                    const int xh = ((k & 3) == 3) ? ((int)a >> 24) : (int)(signed char)((a >> (8 * (k & 3))) & 0xFF);
                    const int xt = ((k & 3) == 3) ? ((int)b >> 24) : (int)(signed char)((b >> (8 * (k & 3))) & 0xFF);
                    const double dh = (double)xh, dt = (double)xt;
                    const double2 Bk = s_car.B[k];
                    Hc = __builtin_fma(dh, Bk.x, Hc);
                    Hs = __builtin_fma(dh, Bk.y, Hs);
                    Tc = __builtin_fma(dt, Bk.x, Tc);
                    Ts = __builtin_fma(dt, Bk.y, Ts);
                }
                // rotate the runs by their start phasors: cos part -> Q, sin part -> I (tracking.py:205-207)
                const double hQ = __builtin_fma(gh.x, Hc, -(gh.y * Hs)), hI = __builtin_fma(gh.y, Hc, gh.x * Hs);
                const double tQ = __builtin_fma(gt.x, Tc, -(gt.y * Ts)), tI = __builtin_fma(gt.y, Tc, gt.x * Ts);
                // code of the tail: switched iff the ramp's switch sample is the run boundary
                const double cEt = (eE <= e2) ? cEn : cEh;
                const double cLt = (eL <= e2) ? cLn : cLh;
                aIE = __builtin_fma(cEt, tI, __builtin_fma(cEh, hI, aIE));
                aQE = __builtin_fma(cEt, tQ, __builtin_fma(cEh, hQ, aQE));
                aIP = __builtin_fma(cP, tI + hI, aIP);
                aQP = __builtin_fma(cP, tQ + hQ, aQP);
                aIL = __builtin_fma(cLt, tI, __builtin_fma(cLh, hI, aIL));
                aQL = __builtin_fma(cLt, tQ, __builtin_fma(cLh, hQ, aQL));
            }
        }
        s_red[0][tid] = aIE;
        s_red[1][tid] = aQE;
        s_red[2][tid] = aIP;
        s_red[3][tid] = aQP;
        s_red[4][tid] = aIL;
        s_red[5][tid] = aQL;
        __syncthreads();
        if (wave < 3) {
            const int v = 2 * wave + (lane >> 5), l = lane & 31;
            double acc = s_red[v][l];
#pragma unroll
            for (int k = 1; k < TRK_THREADS / 32; ++k) acc += s_red[v][l + 32 * k];
            acc = half_wave_sum(acc, lane);
            if (l == 0) s_tot[v] = acc;
        }
        __syncthreads();
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
            const double carrError = div_rn(atan(Q_P / I_P) / 2.0, M_PI, K.inv_pi);
            const double carrNco = oldNco + K.k_carr_a * (carrError - oldErr) + carrError * K.k_carr_b;
            const double carrFreq = basis + carrNco;
            const double w_new = (carrFreq * 2.0) * M_PI;
            if (more) {
                tp_tables(K, w_new, rc, s_car, lane, 0);
                tp_tables(K, w_new, rc, s_car, lane, 1);
            }
            if (lane == 0) {
                s_st.w = w_new;
                s_st.remCarr = rc;
                s_st.oldCarrNco = carrNco;
                s_st.oldCarrErr = carrError;
                s_st.carrFreq = carrFreq;
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
                o[0 * m + it] = (double)(pos_after + K.file_off);
                o[1 * m + it] = codeFreq;
                o[9 * m + it] = codeError;
                o[10 * m + it] = codeNco;
            }
            if (more) prep_code(K, codeFreq, rem_next, pos_after, s_st, s_blk, lane == 0);
        }
        done = it + 1;
        __syncthreads();   // next block's parameters visible
    }
    if (tid == 0) ms_done[ch] = done;
}

void sgx_trk_tp_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                       double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                       int* err) {
    (void)prof;
    (void)xch;
    (void)err;
    trk_kernel_tp<<<K.n_ch, TRK_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K);
    (void)n_blocks;
}
