// Speculative tracking pipeline for gfx950 (reference tracking.py:13-295).
//
// The cooperative kernel in sgx_trk.hip spends a block as  map -> reduce/exchange -> loop filter, strictly
// in sequence, because the map needs the parameters the filter produces.  But of those parameters only the
// two NCO RATES (code and carrier frequency) are new information: the next block's first sample, its code
// phase and its carrier phase follow exactly from the current block's parameters (tracking.py:190,197,255).
// This kernel therefore runs the per-sample work of block k+1 in the shadow of block k's exchange and
// filter, with the rates of block k as a prediction, in a form that can be corrected EXACTLY:
//
//   code     the predicted ramps give, per group of 16 samples, the chip index at the first sample, the
//            switch sample, and a MARGIN: the smallest distance (in chips) between any chip boundary
//            and the samples next to it.  The true ramps differ by at most i*|dstep| (+1e-12), so if
//            that bound is below the margin the chip indices are provably unchanged; otherwise the
//            group is recomputed with the exact arithmetic.  (Typical: |dstep| ~ 1e-9 chips/sample,
//            a handful of groups per block are recomputed.)
//   carrier  the group sums are taken with the previous block's rotation table B_b and kept as three
//            moments  sum b^m x_b B_b, m = 0,1,2.  With the true rate, exp(j b dtheta) =
//            1 + j b dtheta - (b dtheta)^2/2 + O((b dtheta)^3): the blocks where |15 dtheta| >= 8e-5
//            (rate jump above ~30 Hz: pull-in transients) are recomputed exactly, elsewhere the
//            truncation error is below 1e-13 relative.  The group-start phasor is always taken from the
//            true tables.
//
// Wave roles inside a 384-thread workgroup (one workgroup = one of `split` members of a channel, exactly
// one 256-group unit per member): wave 0 CAR = partial-sum exchange + PLL + carrier tables, wave 1 COD =
// DLL + code parameters + prediction + result stores, waves 2..5 MAP.  They synchronise through LDS epoch
// flags (no s_barrier in the steady state); members exchange through tagged granules exactly as in
// sgx_trk.hip.  Every spin is bounded; a timeout sets the error word and the call fails.
#include "sgx_trk_common.h"

#define SPEC_THREADS 384
#define SPEC_MAPW 4
// role timing (SGX_TRK_PROFILE=1): shader cycles accumulated by lane 0 of the CAR, COD and first MAP wave of
// member 0: prof[16*ch + slot]
#define TICK() ((long long)__builtin_amdgcn_s_memtime())
#define LAP(slot)                              \
    do {                                       \
        if (prof) {                            \
            const long long t_ = TICK();       \
            pacc[slot] += t_ - tlast;          \
            tlast = t_;                        \
        }                                      \
    } while (0)

struct SpecCode {            // code side of a block (true parameters, or the prediction for the next one)
    long long pos;
    int blk;
    int stop;
    int blk_diff;            // true block length differs from the predicted one
    int pad;
    double startE, stepE, startP, stepP, startL, stepL, inv_step;
    double dstep_max;        // max |true step - predicted step| over the three ramps
};

struct SpecCarr {            // carrier side of a block
    double dtheta;           // phase increment per sample minus the previous block's (rad)
    double2 B[16], W1[16], W2[16], W3[16];
};

struct CodeVals {            // register copy of the scalar code computation (uniform across the wave)
    long long pos;
    int blk, stop;
    double startE, stepE, startP, stepP, startL, stepL, inv_step, rem_next;
};

// tracking.py:148-190 scalar part (T1, T3, T4) - same arithmetic as prep_code in sgx_trk_common.h
__device__ __forceinline__ CodeVals code_values(const TrkConst& K, double codeFreq, double rem, long long pos) {
    CodeVals v;
    const double step = div_rn(codeFreq, K.fs, K.inv_fs);                    // codeFreq / fs (T1)
    const int blk = (int)ceil((K.code_len - rem) / step);
    const double nb = (double)blk;
    const double span = nb * step;
    const int ki = blk - K.nb_base;
    const bool known = (ki >= 0 && ki < 8);
    const double ynb = known ? K.inv_nb[ki] : 0.0;
    v.pos = pos;
    v.blk = blk;
    v.stop = (blk <= 0 || pos + blk > K.rec_len) ? 1 : 0;
    v.startE = rem - K.spacing;
    const double dE = ((span + rem) - K.spacing) - v.startE;
    v.startL = rem + K.spacing;
    const double dL = ((span + rem) + K.spacing) - v.startL;
    v.startP = rem;
    const double dP = (span + rem) - rem;
    if (known) {   // delta / blk (T3) through the precomputed reciprocal of the block length
        v.stepE = div_rn(dE, nb, ynb);
        v.stepL = div_rn(dL, nb, ynb);
        v.stepP = div_rn(dP, nb, ynb);
    } else {
        v.stepE = dE / nb;
        v.stepL = dL / nb;
        v.stepP = dP / nb;
    }
    const double r0 = __builtin_amdgcn_rcp(step);
    v.inv_step = __builtin_fma(r0, __builtin_fma(-step, r0, 1.0), r0);
    const double t_last = ramp_at(blk - 1, v.stepP, rem);
    v.rem_next = (t_last + step) - 1023.0;
    return v;
}

__device__ __forceinline__ void store_code(SpecCode& d, const CodeVals& v, double dstep_max, int blk_diff) {
    d.pos = v.pos;
    d.blk = v.blk;
    d.stop = v.stop;
    d.blk_diff = blk_diff;
    d.startE = v.startE;
    d.stepE = v.stepE;
    d.startP = v.startP;
    d.stepP = v.stepP;
    d.startL = v.startL;
    d.stepL = v.stepL;
    d.inv_step = v.inv_step;
    d.dstep_max = dstep_max;
}

// carrier tables of a block with rate w, start phase remCarr, `head` bytes before the first sample
__device__ __forceinline__ void carr_tables(const TrkConst& K, double w, double remCarr, int head, SpecCarr& b,
                                            int lane, double& r_hi, double& r_lo) {
    r_hi = w * K.inv_2pifs_hi;
    r_lo = __builtin_fma(w, K.inv_2pifs_hi, -r_hi) + w * K.inv_2pifs_lo;
    const int sel = lane >> 4, idx = lane & 15;
    const double mult = (sel == 0) ? (double)idx
                      : (sel == 1) ? (double)(16 * idx)
                      : (sel == 2) ? (double)(256 * idx)
                                   : (double)(TRK_UNIT * idx - head);
    const double p = r_hi * mult;
    const double e = __builtin_fma(r_hi, mult, -p) + r_lo * mult;
    double u = (p - floor(p)) + e;
    if (sel == 3) {
        u += remCarr * K.inv_2pi;
        u -= (u >= 1.0) ? 1.0 : 0.0;
    }
    double sn, cs;
    sincos_turns(u, sn, cs);
    const double2 v = make_double2(cs, sn);
    if (sel == 0) b.B[idx] = v;
    else if (sel == 1) b.W1[idx] = v;
    else if (sel == 2) b.W2[idx] = v;
    else b.W3[idx] = v;
}

// sum over the 64 lanes of a wave, result in every lane (fixed order)
__device__ __forceinline__ double wave_sum64(double v) {
    v = row_sum(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- LDS epoch flags -------------------------------------------------------------------------------------
template <bool SLEEP = true>
__device__ __forceinline__ bool flag_wait(volatile int* f, int target, int* err, int code) {
    int budget = 1 << 24;
    while (*f < target) {
        if (--budget == 0) {
            atomicExch(err, code);
            return false;
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

__device__ __forceinline__ void flag_set(volatile int* f, int v, bool writer) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (writer) *f = v;
}

__device__ __forceinline__ void mask_bytes(unsigned (&wd)[4], int i0, int blk) {
    if (i0 < 0 || i0 + 16 > blk) {
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
}

// one-rounding ramp estimate (differs from the reference's two-rounding value by < 3e-13)
__device__ __forceinline__ double ramp_est(int i, double step, double start) {
    return __builtin_fma((double)i, step, start);
}

// margin of one ramp inside the group [ilo, ilast]: see the header comment
__device__ __forceinline__ double ramp_margin(double start, double step, int ilo, int ilast, int k1, int isw) {
    const double kd = (double)k1;
    const double t_lo = ramp_est(ilo, step, start);
    double m = fmin(t_lo - (kd - 1.0), kd - t_lo);                 // chip index at ilo stays k1
    const int ib = (isw - 1 < ilast) ? isw - 1 : ilast;             // last sample that must stay at k1
    m = fmin(m, kd - ramp_est(ib, step, start));
    if (isw <= ilast) m = fmin(m, ramp_est(isw, step, start) - kd);   // first sample that must stay above
    return m;
}

__global__ __launch_bounds__(SPEC_THREADS) void trk_spec_kernel(const int8_t* __restrict__ rec,
                                                                const int8_t* __restrict__ codes,
                                                                const TrkChan* __restrict__ chans,
                                                                double* __restrict__ out,
                                                                int* __restrict__ ms_done, TrkConst K,
                                                                long long* __restrict__ prof,
                                                                unsigned long long* __restrict__ xch,
                                                                int* __restrict__ err) {
    __shared__ unsigned s_code_hi[1028];
    __shared__ SpecCode s_code[2];
    __shared__ SpecCode s_pred[2];
    __shared__ SpecCarr s_carr[2];
    __shared__ double s_part[2][16][8];    // [parity][row of 16 map lanes][value]
    __shared__ double s_tot[2][6];
    __shared__ int s_flag[8];   // 0 f_code, 1 f_carr, 2 f_pred, 3 f_tot, 4 f_partcnt, 5 same-XCD

    volatile int* f_code = &s_flag[0];
    volatile int* f_carr = &s_flag[1];
    volatile int* f_pred = &s_flag[2];
    volatile int* f_tot = &s_flag[3];
    volatile int* f_part = &s_flag[4];

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
    unsigned long long* __restrict__ xbase = xch + (long long)ch * (2 * TRK_MAX_SPLIT * 12 + 16);
    const int ecode = 1 + ch;

    for (int i = tid; i < 1028; i += SPEC_THREADS) {
        int j = i - 1;
        if (j < 0) j = 1022;
        if (j >= 1023) j -= 1023;
        if (j >= 1023) j -= 1023;
        s_code_hi[i] = (codes[(cc.prn - 1) * 1023 + j] > 0) ? 0x3FF00000u : 0xBFF00000u;
    }
    if (tid < 8) s_flag[tid] = 0;
    __syncthreads();

    // ---- placement check through the placement-independent path (same as sgx_trk.hip) ------------------
    bool fast = false;
    if (P > 1) {
        unsigned long long* pl = xbase + 2 * TRK_MAX_SPLIT * 12;
        const unsigned me = xcc_id();
        if (tid == 0) __hip_atomic_store(pl + member, 0xC0DE000000000000ull | me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            unsigned long long x = 0;
            int budget = 1 << 22;
            for (;;) {
                if (lane < P) x = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = lane >= P || (x >> 48) == 0xC0DE;
                if (__all(ok)) break;
                if (--budget == 0) {
                    if (lane == 0) atomicExch(err, ecode);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool same = __all(lane >= P || (unsigned)(x & 0xF) == me);
            if (lane == 0) s_flag[5] = same ? 1 : 0;
        }
        __syncthreads();
        fast = (s_flag[5] != 0) && (K.fast_xcd != 0);
    }

    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    const long long m = K.ms;
    const double two_pi = 2 * M_PI;
    const long long limit = K.rec_alloc - 16;

    if (wave == 0) {
        // =============================== CAR: exchange + PLL + carrier tables ===============================
        __builtin_amdgcn_s_setprio(3);
        double w = (cc.acquiredFreq * 2.0) * M_PI, remCarr = 0.0;
        double oldNco = 0.0, oldErr = 0.0;
        const double basis = cc.acquiredFreq;
        double r_hi, r_lo;
        carr_tables(K, w, remCarr, (int)(cc.pos0 & 15), s_carr[0], lane, r_hi, r_lo);
        if (lane == 0) s_carr[0].dtheta = 0.0;
        flag_set(f_carr, 1, lane == 0);
        long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = prof ? TICK() : 0;
        for (int it = 0; it < K.ms; ++it) {
            const int p = it & 1;
            if (!flag_wait(f_code, it + 1, err, ecode)) break;
            const int blk = s_code[p].blk;
            const long long pos = s_code[p].pos;
            if (s_code[p].stop) break;
            // T5: carrier phase at the end of this block, exact remainder by one FMA (off the critical path)
            const double arg_end = w * ((double)blk / K.fs) + remCarr;
            const double kq = floor(arg_end * K.inv_2pi);
            double rc = __builtin_fma(-kq, two_pi, arg_end);
            if (rc < 0.0) rc += two_pi;
            if (rc >= two_pi) rc -= two_pi;
            LAP(0);   // CAR: end-phase bookkeeping (+ wait for the code flag)
            if (!flag_wait<false>(f_part, SPEC_MAPW * (it + 1), err, ecode)) break;
            LAP(1);   // CAR: waiting for the map waves' partials
            // 16 row partials per value -> member sums: lane = 16*q + r reads row r of value q (and q+4 for q < 2)
            const int row = lane >> 4, c = lane & 15;
            const double tA = row_sum(s_part[p][c][row]);
            const double tB = row_sum(row < 2 ? s_part[p][c][row + 4] : 0.0);
            if (P > 1) {
                const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
                if (c == 0) {
                    unsigned long long* gp = xbase + ((it & 1) * TRK_MAX_SPLIT + member) * 12 + 2 * row;
                    granule_store(gp, tag | (unsigned)__double2loint(tA), fast);
                    granule_store(gp + 1, tag | (unsigned)__double2hiint(tA), fast);
                    if (row < 2) {
                        granule_store(gp + 8, tag | (unsigned)__double2loint(tB), fast);
                        granule_store(gp + 9, tag | (unsigned)__double2hiint(tB), fast);
                    }
                }
                LAP(4);   // CAR: partial row sums + publish
                const unsigned epoch = (unsigned)(it + 1);
                const bool mA = c < P, mB = mA && row < 2;
                const unsigned long long* gA = xbase + ((it & 1) * TRK_MAX_SPLIT + c) * 12 + 2 * row;
                const unsigned long long* gB = gA + 8;
                unsigned long long a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                int budget = 1 << 22;
                for (;;) {
                    if (mA) {
                        a0 = __hip_atomic_load(gA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a1 = __hip_atomic_load(gA + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (mB) {
                        b0 = __hip_atomic_load(gB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        b1 = __hip_atomic_load(gB + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const bool ok = (!mA || ((unsigned)(a0 >> 32) == epoch && (unsigned)(a1 >> 32) == epoch)) &&
                                    (!mB || ((unsigned)(b0 >> 32) == epoch && (unsigned)(b1 >> 32) == epoch));
                    if (__all(ok)) break;
                    if (--budget == 0) {
                        if (lane == 0) atomicExch(err, ecode);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                LAP(5);   // CAR: polling until every member's granules are visible
                const double dA = mA ? __hiloint2double((int)(unsigned)a1, (int)(unsigned)a0) : 0.0;
                const double dB = mB ? __hiloint2double((int)(unsigned)b1, (int)(unsigned)b0) : 0.0;
                const double sA = row_sum(dA), sB = row_sum(dB);
                if (c == 0) {
                    s_tot[p][row] = sA;
                    if (row < 2) s_tot[p][row + 4] = sB;
                }
            } else if (c == 0) {
                s_tot[p][row] = tA;
                if (row < 2) s_tot[p][row + 4] = tB;
            }
            flag_set(f_tot, it + 1, lane == 0);
            LAP(2);   // CAR: publish + gather + totals
            // T7 PLL (tracking.py:223-235)
            const double I_P = s_tot[p][2], Q_P = s_tot[p][3];
            const double carrError = div_rn(atan(Q_P / I_P) / 2.0, M_PI, K.inv_pi);   // atan(Q/I) / 2 / pi
            const double carrNco = oldNco + K.k_carr_a * (carrError - oldErr) + carrError * K.k_carr_b;
            const double carrFreq = basis + carrNco;
            const double w_new = (carrFreq * 2.0) * M_PI;
            LAP(6);   // CAR: PLL
            if (it + 1 < K.ms) {
                double n_hi, n_lo;
                carr_tables(K, w_new, rc, (int)((pos + blk) & 15), s_carr[p ^ 1], lane, n_hi, n_lo);
                if (lane == 0) s_carr[p ^ 1].dtheta = ((n_hi - r_hi) + (n_lo - r_lo)) * two_pi;
                r_hi = n_hi;
                r_lo = n_lo;
                flag_set(f_carr, it + 2, lane == 0);
            }
            if (lane == 0 && member == 0) {   // T9 record, after the flags: off the critical path
                o[2 * m + it] = carrFreq;
                o[11 * m + it] = carrError;
                o[12 * m + it] = carrNco;
            }
            w = w_new;
            remCarr = rc;
            oldNco = carrNco;
            oldErr = carrError;
            LAP(3);   // CAR: PLL + tables + stores
        }
        if (prof && lane == 0 && (member == 0 || member == 5))
            for (int k = 0; k < 8; ++k) prof[ch * 64 + (member == 0 ? 0 : 8) + k] = pacc[k];
    } else if (wave == 1) {
        // =============================== COD: DLL + code parameters + prediction ===========================
        __builtin_amdgcn_s_setprio(3);
        double oldNco = 0.0, oldErr = 0.0;
        CodeVals cur = code_values(K, K.code_basis, 0.0, cc.pos0);
        if (lane == 0) store_code(s_code[0], cur, 0.0, 0);
        flag_set(f_code, 1, lane == 0);
        CodeVals prd = code_values(K, K.code_basis, cur.rem_next, cur.pos + cur.blk);
        if (lane == 0) store_code(s_pred[0], prd, 0.0, 0);
        flag_set(f_pred, 1, lane == 0);
        int done = 0;
        long long pacc[3] = {0, 0, 0};
        long long tlast = prof ? TICK() : 0;
        for (int it = 0; it < K.ms; ++it) {
            const int p = it & 1;
            if (cur.stop) break;
            if (!flag_wait<false>(f_tot, it + 1, err, ecode)) break;
            LAP(0);   // COD: waiting for the totals
            // T8 DLL (tracking.py:238-251)
            const double I_E = s_tot[p][0], Q_E = s_tot[p][1], I_L = s_tot[p][4], Q_L = s_tot[p][5];
            const double eE = sqrt(I_E * I_E + Q_E * Q_E);
            const double eL = sqrt(I_L * I_L + Q_L * Q_L);
            const double codeError = (eE - eL) / (eE + eL);
            const double codeNco = oldNco + K.k_code_a * (codeError - oldErr) + codeError * K.k_code_b;
            const double codeFreq = K.code_basis - codeNco;
            const long long pos_after = cur.pos + cur.blk;
            CodeVals nxt = cur;
            if (it + 1 < K.ms) {
                nxt = code_values(K, codeFreq, cur.rem_next, pos_after);
                const double ds = fmax(fmax(fabs(nxt.stepE - prd.stepE), fabs(nxt.stepP - prd.stepP)),
                                       fabs(nxt.stepL - prd.stepL));
                if (lane == 0) store_code(s_code[p ^ 1], nxt, ds, nxt.blk != prd.blk ? 1 : 0);
                flag_set(f_code, it + 2, lane == 0);
                LAP(2);   // COD: DLL + true parameters (critical part)
                prd = code_values(K, codeFreq, nxt.rem_next, nxt.pos + nxt.blk);
                if (lane == 0) store_code(s_pred[p ^ 1], prd, 0.0, 0);
                flag_set(f_pred, it + 2, lane == 0);
            }
            if (member == 0) {   // T9 record (tracking.py:255-275)
                if (lane == 0) {
                    o[0 * m + it] = (double)(pos_after + K.file_off);
                    o[1 * m + it] = codeFreq;
                    o[9 * m + it] = codeError;
                    o[10 * m + it] = codeNco;
                }
                if (lane >= 8 && lane < 14) {
                    const int k = lane - 8;   // s_tot order: I_E Q_E I_P Q_P I_L Q_L
                    const int series = (k == 0) ? 4 : (k == 1) ? 6 : (k == 2) ? 3 : (k == 3) ? 7 : (k == 4) ? 5 : 8;
                    o[series * m + it] = s_tot[p][k];
                }
            }
            oldNco = codeNco;
            oldErr = codeError;
            cur = nxt;
            done = it + 1;
            LAP(1);   // COD: DLL + parameters + prediction + stores
        }
        if (prof && lane == 0 && member == 0)
            for (int k = 0; k < 3; ++k) prof[ch * 64 + 16 + k] = pacc[k];
        if (lane == 0 && member == 0) ms_done[ch] = done;
    } else {
        // =============================== MAP: finalize block it, shadow block it+1 ==========================
        const int mw = wave - 2;
        const int ml = tid - 128;                 // 0..255: the lane's group inside the unit
        const int g = member * TRK_THREADS + ml;  // group index inside a block
        const long long lane_off = (long long)g * 16;
        uint4 raw = load_group(rec, (cc.pos0 & ~15ll) + lane_off, limit);
        // shadow state of the block about to be finalized
        bool have = false;
        double M0c = 0, M0s = 0, M1c = 0, M1s = 0, M2c = 0, M2s = 0, T0c = 0, T0s = 0, T1c = 0, T1s = 0, T2c = 0, T2s = 0;
        double margin = 0.0;
        int skE = 0, skP = 0, skL = 0, sflags = 0;
        long long n_exact = 0;
        long long pacc[5] = {0, 0, 0, 0, 0};
        long long tlast = prof ? TICK() : 0;
        for (int it = 0; it < K.ms; ++it) {
            const int p = it & 1;
            // everything the finalize can know before the true parameters arrive: the code signs at the
            // shadow's chip indices (constant table), read while this wave would otherwise idle
            const double cE1 = __hiloint2double((int)s_code_hi[skE], 0), cE2 = __hiloint2double((int)s_code_hi[skE + 1], 0);
            const double cP1 = __hiloint2double((int)s_code_hi[skP], 0), cP2 = __hiloint2double((int)s_code_hi[skP + 1], 0);
            const double cL1 = __hiloint2double((int)s_code_hi[skL], 0), cL2 = __hiloint2double((int)s_code_hi[skL + 1], 0);
            const double dE = (sflags & 1) ? (cE2 - cE1) : 0.0;
            const double dP = (sflags & 2) ? (cP2 - cP1) : 0.0;
            const double dL = (sflags & 4) ? (cL2 - cL1) : 0.0;
            if (!flag_wait(f_code, it + 1, err, ecode)) break;
            if (!flag_wait(f_carr, it + 1, err, ecode)) break;
            LAP(0);   // MAP: waiting for the block's true parameters
            const SpecCode& C = s_code[p];
            const SpecCarr& R = s_carr[p];
            // one batch of LDS reads
            const long long pos = C.pos;
            const int blk = C.blk;
            const int stop = C.stop, blk_diff = C.blk_diff;
            const double dstep_max = C.dstep_max;
            const double dth = R.dtheta;
            const double2 a1 = R.W1[ml & 15], a2 = R.W2[ml >> 4], a3 = R.W3[member];
            if (stop) break;
            const int head = (int)(pos & 15);
            const int n_groups = (head + blk + 15) >> 4;
            const int i0 = g * 16 - head;
            const bool valid = g < n_groups;

            // group-start phasor from the true tables
            const double lc = __builtin_fma(a1.x, a2.x, -(a1.y * a2.y));
            const double ls = __builtin_fma(a1.x, a2.y, a1.y * a2.x);
            const double gc = __builtin_fma(lc, a3.x, -(ls * a3.y));
            const double gs = __builtin_fma(lc, a3.y, ls * a3.x);

            const bool blk_bad = !have || fabs(dth) * 15.0 >= 8e-5;
            const double bound = (double)(i0 + 16) * dstep_max + 1e-12;
            const bool near_end = blk_diff && (i0 + 17 >= blk - 1);
            const bool lane_bad = valid && ((sflags & 8) || !(margin > bound) || near_end);
            double vIE = 0.0, vQE = 0.0, vIP = 0.0, vQP = 0.0, vIL = 0.0, vQL = 0.0;
            if (blk_bad || __any(lane_bad)) {
                // ---- exact path for this wave: the reference's chip indices and carrier from the true tables
                ++n_exact;
                if (valid) {
                    unsigned wd[4] = {raw.x, raw.y, raw.z, raw.w};
                    mask_bytes(wd, i0, blk);
                    const int ilo = i0 < 0 ? 0 : i0;
                    int kE, swE, kP, swP, kL, swL;
                    ramp_setup(C.startE, C.stepE, C.inv_step, ilo, kE, swE);
                    ramp_setup(C.startP, C.stepP, C.inv_step, ilo, kP, swP);
                    ramp_setup(C.startL, C.stepL, C.inv_step, ilo, kL, swL);
                    const double xcE1 = __hiloint2double((int)s_code_hi[kE], 0), xcE2 = __hiloint2double((int)s_code_hi[kE + 1], 0);
                    const double xcP1 = __hiloint2double((int)s_code_hi[kP], 0), xcP2 = __hiloint2double((int)s_code_hi[kP + 1], 0);
                    const double xcL1 = __hiloint2double((int)s_code_hi[kL], 0), xcL2 = __hiloint2double((int)s_code_hi[kL + 1], 0);
                    unsigned w0 = wd[0], w1 = wd[1], w2 = wd[2], w3 = wd[3];
#pragma unroll 1
                    for (int b = 0; b < 16; ++b) {
                        const int i = i0 + b;
                        const double xd = (double)(int)(signed char)(w0 & 0xFF);
                        w0 = (w0 >> 8) | (w1 << 24);
                        w1 = (w1 >> 8) | (w2 << 24);
                        w2 = (w2 >> 8) | (w3 << 24);
                        w3 >>= 8;
                        const double2 Bb = R.B[b];
                        const double c = __builtin_fma(gc, Bb.x, -(gs * Bb.y));
                        const double s = __builtin_fma(gs, Bb.x, gc * Bb.y);
                        const double xs = s * xd, xc = c * xd;
                        const double cE = i >= swE ? xcE2 : xcE1;
                        const double cP = i >= swP ? xcP2 : xcP1;
                        const double cL = i >= swL ? xcL2 : xcL1;
                        vIE = __builtin_fma(cE, xs, vIE);
                        vQE = __builtin_fma(cE, xc, vQE);
                        vIP = __builtin_fma(cP, xs, vIP);
                        vQP = __builtin_fma(cP, xc, vQP);
                        vIL = __builtin_fma(cL, xs, vIL);
                        vQL = __builtin_fma(cL, xc, vQL);
                    }
                }
            } else if (valid) {
                // ---- fast finalize: second-order rate correction of the moments, true phasor, code signs
                const double h = 0.5 * dth * dth;
                const double Ac = __builtin_fma(-h, M2c, __builtin_fma(-dth, M1s, M0c));
                const double As = __builtin_fma(-h, M2s, __builtin_fma(dth, M1c, M0s));
                const double Tc = __builtin_fma(-h, T2c, __builtin_fma(-dth, T1s, T0c));
                const double Ts = __builtin_fma(-h, T2s, __builtin_fma(dth, T1c, T0s));
                const double allQ = __builtin_fma(gc, Ac, -(gs * As));
                const double allI = __builtin_fma(gs, Ac, gc * As);
                const double tlQ = __builtin_fma(gc, Tc, -(gs * Ts));
                const double tlI = __builtin_fma(gs, Tc, gc * Ts);
                vIE = __builtin_fma(dE, tlI, cE1 * allI);
                vQE = __builtin_fma(dE, tlQ, cE1 * allQ);
                vIP = __builtin_fma(dP, tlI, cP1 * allI);
                vQP = __builtin_fma(dP, tlQ, cP1 * allQ);
                vIL = __builtin_fma(dL, tlI, cL1 * allI);
                vQL = __builtin_fma(dL, tlQ, cL1 * allQ);
            }
            LAP(1);   // MAP: finalize
            // ---- wave sums -> LDS partials -> counter
            vIE = row_sum(vIE);
            vQE = row_sum(vQE);
            vIP = row_sum(vIP);
            vQP = row_sum(vQP);
            vIL = row_sum(vIL);
            vQL = row_sum(vQL);
            if ((lane & 15) == 0) {
                double* sp = s_part[p][mw * 4 + (lane >> 4)];   // s_tot order: I_E Q_E I_P Q_P I_L Q_L
                sp[0] = vIE;
                sp[1] = vQE;
                sp[2] = vIP;
                sp[3] = vQP;
                sp[4] = vIL;
                sp[5] = vQL;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd((int*)f_part, 1);
            LAP(2);   // MAP: wave sums + partials

            // the next block starts at pos + blk: fetch this lane's group of it (the wave has slack now)
            raw = load_group(rec, ((pos + blk) & ~15ll) + lane_off, limit);
            if (it + 1 >= K.ms) break;
            // ---- shadow of block it+1 with the predicted code parameters and this block's rotation table
            if (!flag_wait(f_pred, it + 1, err, ecode)) break;
            LAP(3);   // MAP: waiting for the prediction
            const SpecCode& Q = s_pred[p];
            {
                const int head2 = (int)(Q.pos & 15);
                const int j0 = g * 16 - head2;
                const int blk2 = Q.blk;
                unsigned wd[4] = {raw.x, raw.y, raw.z, raw.w};
                mask_bytes(wd, j0, blk2);
                const int ilo = j0 < 0 ? 0 : j0;
                const int ilast = (j0 + 15 < blk2 - 1) ? j0 + 15 : blk2 - 1;
                int kE, swE, kP, swP, kL, swL;
                ramp_setup(Q.startE, Q.stepE, Q.inv_step, ilo, kE, swE);
                ramp_setup(Q.startP, Q.stepP, Q.inv_step, ilo, kP, swP);
                ramp_setup(Q.startL, Q.stepL, Q.inv_step, ilo, kL, swL);
                double mg = ramp_margin(Q.startE, Q.stepE, ilo, ilast, kE, swE);
                mg = fmin(mg, ramp_margin(Q.startP, Q.stepP, ilo, ilast, kP, swP));
                mg = fmin(mg, ramp_margin(Q.startL, Q.stepL, ilo, ilast, kL, swL));
                const int iend = j0 + 16;
                int swmin = swE < swP ? swE : swP;
                swmin = swL < swmin ? swL : swmin;
                const bool eS = (swE == swmin), pS = (swP == swmin), lS = (swL == swmin);
                const bool odd = (swE < iend && !eS) || (swP < iend && !pS) || (swL < iend && !lS);
                const bool dead = (ilo > ilast);   // no sample of the predicted block in this group
                sflags = (eS ? 1 : 0) | (pS ? 2 : 0) | (lS ? 4 : 0) | ((odd || dead) ? 8 : 0);
                skE = kE < 0 ? 0 : (kE > 1025 ? 1025 : kE);
                skP = kP < 0 ? 0 : (kP > 1025 ? 1025 : kP);
                skL = kL < 0 ? 0 : (kL > 1025 ? 1025 : kL);
                margin = mg;
                const int bsw = swmin - j0;
                double2 B[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) B[b] = R.B[b];
                M0c = M0s = M1c = M1s = M2c = M2s = 0.0;
                T0c = T0s = T1c = T1s = T2c = T2s = 0.0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const unsigned wv = wd[b >> 2];
                    const int xi = ((b & 3) == 3) ? ((int)wv >> 24) : (int)(signed char)((wv >> (8 * (b & 3))) & 0xFF);
                    const double x0 = (double)xi;
                    const double x1 = x0 * (double)b;
                    const double x2 = x1 * (double)b;
                    M0c = __builtin_fma(x0, B[b].x, M0c);
                    M0s = __builtin_fma(x0, B[b].y, M0s);
                    M1c = __builtin_fma(x1, B[b].x, M1c);
                    M1s = __builtin_fma(x1, B[b].y, M1s);
                    M2c = __builtin_fma(x2, B[b].x, M2c);
                    M2s = __builtin_fma(x2, B[b].y, M2s);
                    const bool tl = (b >= bsw);
                    const double t0 = tl ? x0 : 0.0, t1 = tl ? x1 : 0.0, t2 = tl ? x2 : 0.0;
                    T0c = __builtin_fma(t0, B[b].x, T0c);
                    T0s = __builtin_fma(t0, B[b].y, T0s);
                    T1c = __builtin_fma(t1, B[b].x, T1c);
                    T1s = __builtin_fma(t1, B[b].y, T1s);
                    T2c = __builtin_fma(t2, B[b].x, T2c);
                    T2s = __builtin_fma(t2, B[b].y, T2s);
                }
                have = true;
            }
            LAP(4);   // MAP: shadow
        }
        if (prof && lane == 0 && member == 0) {
            atomicAdd((unsigned long long*)&prof[ch * 64 + 63], (unsigned long long)n_exact);
            if (mw == 0)
                for (int k = 0; k < 5; ++k) prof[ch * 64 + 24 + k] = pacc[k];
            if (mw == 2)
                for (int k = 0; k < 5; ++k) prof[ch * 64 + 32 + k] = pacc[k];
        }
    }
}

// host launcher, called by sgx_track (sgx_trk.hip)
void sgx_trk_spec_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                         double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                         int* err) {
    trk_spec_kernel<<<n_blocks, SPEC_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K, prof, xch,
                                                       err);
}
