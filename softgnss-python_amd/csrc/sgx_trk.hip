// TrackingResult.track on gfx950 (reference tracking.py:13-295).
//
// A channel is a chain of `ms` dependent 1-ms steps: every block's length, code ramps and NCO rates
// depend on the previous block's six correlator sums.  Channels are independent.  The kernel is
// persistent: one launch walks all code periods of all channels.
//
// Work decomposition
//   * a block (~38 192 samples) is cut into UNITS of 256 groups x 16 samples (4 KiB of IF);
//   * `split` workgroups of 256 threads (4 waves, one per SIMD of a CU) cooperate on one channel,
//     member c takes units c, c+split, ...; split = 1 keeps a channel on one CU (throughput mode,
//     many channels), split = 10 spreads a channel over ten CUs of one XCD (few channels, latency mode);
//   * members exchange their six partial sums once per block through HBM/L2 as tagged 8-byte granules
//     (write-through stores, relaxed agent-scope polls, epoch tags, double-buffered by epoch parity,
//     bounded spins) and then every member runs the loop filter redundantly, so the next block's
//     parameters need no broadcast.  Nothing depends on which CU or XCD a member lands on.
//
// Per block (one loop iteration)
//   map     every lane takes 16 consecutive int8 samples of a unit as ONE aligned 16-byte load (a wave
//           reads 1 KiB contiguous); the load of the lane's next unit (or the next block's first unit)
//           is issued before the current one is processed, so HBM latency hides behind arithmetic.
//           * code replicas: the three linspace ramps t = fl(fl(i*step)+start) are monotonic and move
//             0.43 chip over 16 samples, so a group holds at most ONE chip switch (prompt at integer t,
//             early/late together at half-integer t).  Chip index at the group's first sample and the
//             switch sample come from the exact reference arithmetic (an estimate plus two exact
//             probes), which keeps the indices bit-identical to code[int64(ceil(linspace(...)))].
//           * carrier: sample b of a group has phasor G*B_b, G = the lane's group-start phasor (fp64
//             "turns" reduction + one sincospi per block, then a rotation per further unit) and
//             B_b = exp(j b delta), a 16-entry per-block table held in registers.  The lane accumulates
//             sum x_b B_b over the group and over the samples after the switch (5 fp64 ops per sample),
//             then applies G and the code signs once per group.
//           * a group in which early and late switch at different samples (fp64 rounding at a chip
//             boundary) falls back to an exact per-sample loop.
//   reduce  six fp64 partials per lane -> LDS transpose -> 3 waves fold them (DPP) -> exchange.
//   filter  wave 0 runs the PLL and prepares the carrier parameters while wave 1 runs the DLL and
//           prepares block size and code ramps, both with the reference's fp64 operation order
//           (-ffp-contract=off; fused multiply-adds only where written as __builtin_fma).
// fp64 everywhere: 1e-7 errors in the sums move the code NCO enough to flip a chip-boundary sample
// somewhere in a 37 s run, which is a 1e-3 relative blip.
#include "sgx_trk_common.h"

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
    __shared__ double s_rc;                // carrier phase at the end of the current block (wave 3 -> wave 0)
    __shared__ double s_out[2][8];         // scalar outputs of a block, staged for wave 2 to store one block later

    // optional phase profile (SGX_TRK_PROFILE=1): shader cycles of lane 0 in map / wait / reduce / filter
    long long pf_map = 0, pf_wait = 0, pf_red = 0, pf_flt = 0;
#ifdef TRK_FINEPROF
    long long fp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long fp_last = 0;
#endif

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
    // exchange area of the channel: [2 epoch parities][TRK_MAX_SPLIT members][12 granules], then one
    // placement granule per member
    unsigned long long* __restrict__ xbase = xch + (long long)ch * (2 * TRK_MAX_SPLIT * 12 + 16);
    bool fast = false;
    if (P > 1) {
        // placement check through the placement-independent path: all members on one XCD?
        unsigned long long* pl = xbase + 2 * TRK_MAX_SPLIT * 12;
        const unsigned me = xcc_id();
        if (tid == 0) __hip_atomic_store(pl + member, 0xC0DE000000000000ull | me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        if (wave == 0) {
            unsigned long long x = 0;
            int budget = 1 << 22;
            for (;;) {
                if (lane < P) x = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = lane >= P || (x >> 48) == 0xC0DE;
                if (__all(ok)) break;
                if (--budget == 0) {
                    if (lane == 0) atomicExch(err, 1 + ch);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            same = __all(lane >= P || (unsigned)(x & 0xF) == me);
            if (lane == 0) s_tot[0] = same ? 1.0 : 0.0;
        }
        __syncthreads();
        fast = (s_tot[0] != 0.0) && (K.fast_xcd != 0);
        __syncthreads();
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
    if (wave == 0) prep_carr(K, s_st.w, s_st.remCarr, (int)(cc.pos0 & 15), s_blk, lane);
    if (wave == 1) prep_code(K, s_st.codeFreq, s_st.remCode, s_st.pos, s_st, s_blk, lane == 0);
    __syncthreads();

    const long long limit = K.rec_alloc - 16;
    const long long lane_off = (long long)(tid + member * TRK_THREADS) * 16;   // byte offset of the lane's first unit
    uint4 cur = load_group(rec, (s_blk.pos & ~15ll) + lane_off, limit);
    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    const double two_pi = 2 * M_PI;
    int done = 0;
    for (int it = 0; it < K.ms; ++it) {
        const long long tk0 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#ifdef TRK_FINEPROF
        fp_last = (long long)__builtin_amdgcn_s_memtime();
#endif
        const long long pos = s_blk.pos;
        const int blk = s_blk.blk;
        if (s_blk.stop) break;   // short read: tracking.py:159-163
        const double startE = s_blk.startE, stepE = s_blk.stepE;
        const double startP = s_blk.startP, stepP = s_blk.stepP;
        const double startL = s_blk.startL, stepL = s_blk.stepL;
        const double inv_step = s_blk.inv_step;
        double2 B[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) B[b] = s_blk.B[b];

        const long long abase = pos & ~15ll;
        const long long abase_next = (pos + blk) & ~15ll;
        const int head = (int)(pos - abase);              // bytes of the first group before the block
        const int n_groups = (head + blk + 15) >> 4;
        double aIE = 0.0, aQE = 0.0, aIP = 0.0, aQP = 0.0, aIL = 0.0, aQL = 0.0;
        // carrier phasor of the lane's group inside a unit: W1[tid & 15] * W2[tid >> 4]
        double lc, ls;
        {
            const double2 a = s_blk.W1[tid & 15], c2 = s_blk.W2[tid >> 4];
            lc = __builtin_fma(a.x, c2.x, -(a.y * c2.y));
            ls = __builtin_fma(a.x, c2.y, a.y * c2.x);
        }

#pragma unroll 1
        for (int u = member; u < K.n_units; u += P) {
            // issue the load of the lane's next unit (this block's, or the first one of the next block)
            const int un = u + P;
            const uint4 nxt = (un < K.n_units) ? load_group(rec, abase + (long long)(tid + un * TRK_THREADS) * 16, limit)
                                               : load_group(rec, abase_next + lane_off, limit);
            const int g = tid + u * TRK_THREADS;
            PROBE(0);   // block parameters, B table, lane phasor, next-unit load issued, current unit landed
            if (g < n_groups) {
                const int i0 = g * 16 - head;             // sample index of byte 0 of this group
                unsigned wd[4] = {cur.x, cur.y, cur.z, cur.w};
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
                // group-start phasor G = (lane part) * W3[u]
                const double2 w3 = s_blk.W3[u];
                const double gc = __builtin_fma(lc, w3.x, -(ls * w3.y));
                const double gs = __builtin_fma(lc, w3.y, ls * w3.x);
                const int ilo = i0 < 0 ? 0 : i0;
                int kE, swE, kP, swP, kL, swL;
                ramp_setup(startE, stepE, inv_step, ilo, kE, swE);
                ramp_setup(startP, stepP, inv_step, ilo, kP, swP);
                ramp_setup(startL, stepL, inv_step, ilo, kL, swL);
                PROBE(1);   // masks, group phasor, three ramp setups
                const double cE1 = __hiloint2double((int)s_code_hi[kE], 0), cE2 = __hiloint2double((int)s_code_hi[kE + 1], 0);
                const double cP1 = __hiloint2double((int)s_code_hi[kP], 0), cP2 = __hiloint2double((int)s_code_hi[kP + 1], 0);
                const double cL1 = __hiloint2double((int)s_code_hi[kL], 0), cL2 = __hiloint2double((int)s_code_hi[kL + 1], 0);
                PROBE(2);   // code lookups
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
                        const double2 Bb = s_blk.B[b];
                        const double c = __builtin_fma(gc, Bb.x, -(gs * Bb.y));
                        const double s = __builtin_fma(gs, Bb.x, gc * Bb.y);
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
                        Ac = __builtin_fma(xd, B[b].x, Ac);
                        As = __builtin_fma(xd, B[b].y, As);
                        const double xt = (b >= bsw) ? xd : 0.0;
                        Tc = __builtin_fma(xt, B[b].x, Tc);
                        Ts = __builtin_fma(xt, B[b].y, Ts);
                    }
                    PROBE(3);   // 16-sample accumulation
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
            cur = nxt;
            PROBE(4);   // group finalisation
        }
        const long long tk1 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        s_red[0][tid] = aIE;
        s_red[1][tid] = aQE;
        s_red[2][tid] = aIP;
        s_red[3][tid] = aQP;
        s_red[4][tid] = aIL;
        s_red[5][tid] = aQL;
        // carrier phase at the end of this block (T5) does not need the sums: the otherwise idle wave 3
        // computes it.  remCarrPhase = trigarg[blk] % (2 pi), trigarg = w*(blk/fs) + rem; exact remainder by FMA
        if (wave == 3) {
            const double arg_end = s_st.w * ((double)blk / K.fs) + s_st.remCarr;
            const double kq = floor(arg_end * K.inv_2pi);
            double rc = __builtin_fma(-kq, two_pi, arg_end);
            if (rc < 0.0) rc += two_pi;
            if (rc >= two_pi) rc -= two_pi;
            if (lane == 0) s_rc = rc;
        }
        PROBE(5);   // partials to LDS, end-of-block carrier phase
        __syncthreads();
        PROBE(6);   // barrier 1 (waits for the slowest wave)
        const long long tk2 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (wave < 3) {
            // wave w folds values 2w (lanes 0..31) and 2w+1 (lanes 32..63): 8 partials per lane, then DPP
            const int v = 2 * wave + (lane >> 5), l = lane & 31;
            double acc = s_red[v][l];
#pragma unroll
            for (int k = 1; k < TRK_THREADS / 32; ++k) acc += s_red[v][l + 32 * k];
            acc = half_wave_sum(acc, lane);
            PROBE(7);   // local fold
            if (P == 1) {
                if (l == 0) s_tot[v] = acc;
            } else if (l == 0) {
                // publish this member's partial as two {epoch, 32-bit payload} granules: ONE aligned 8-byte
                // store each, so a reader never sees a torn granule (Guideline 16, R2)
                const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
                unsigned long long* gp = xbase + ((it & 1) * TRK_MAX_SPLIT + member) * 12 + 2 * v;
                granule_store(gp, tag | (unsigned)__double2loint(acc), fast);
                granule_store(gp + 1, tag | (unsigned)__double2hiint(acc), fast);
            }
            if (P > 1 && wave == 0) {
                // gather every member's granules of this epoch (relaxed, L1-bypassing polls, bounded).
                // lane = 16*row + c reads member c's value `row` (and value row+4 in rows 0, 1)
                const unsigned epoch = (unsigned)(it + 1);
                const int row = lane >> 4, c = lane & 15;
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
                        if (lane == 0) atomicExch(err, 1 + ch);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                PROBE(8);   // publish + gather (includes waiting for the slowest member)
                // row sums over the members, same lane layout and order in every member => identical totals
                const double dA = mA ? __hiloint2double((int)(unsigned)a1, (int)(unsigned)a0) : 0.0;
                const double dB = mB ? __hiloint2double((int)(unsigned)b1, (int)(unsigned)b0) : 0.0;
                const double sA = row_sum(dA), sB = row_sum(dB);
                if (c == 0) {
                    s_tot[row] = sA;
                    if (row < 2) s_tot[row + 4] = sB;
                }
            }
        }
        PROBE(9);   // totals
        __syncthreads();
        PROBE(10);  // barrier 2
        const long long tk3 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const long long m = K.ms;
        const bool more = (it + 1 < K.ms);
        if (wave == 0) {
            // T7 PLL (tracking.py:223-235); carrier parameters of the next block
            const double I_P = s_tot[2], Q_P = s_tot[3];
            const double oldNco = s_st.oldCarrNco, oldErr = s_st.oldCarrErr, basis = s_st.carrBasis;
            const double carrError = div_rn(atan(Q_P / I_P) / 2.0, M_PI, K.inv_pi);   // atan(Q/I) / 2 / pi
            const double carrNco = oldNco + K.k_carr_a * (carrError - oldErr) + carrError * K.k_carr_b;
            const double carrFreq = basis + carrNco;
            const double w_new = (carrFreq * 2.0) * M_PI;
            const double rc = s_rc;
            if (more) prep_carr(K, w_new, rc, (int)((pos + blk) & 15), s_blk, lane);
            if (lane == 0) {
                s_st.w = w_new;
                s_st.remCarr = rc;
                s_st.oldCarrNco = carrNco;
                s_st.oldCarrErr = carrError;
                s_st.carrFreq = carrFreq;
            }
            if (lane == 0 && member == 0) {
                s_out[it & 1][0] = carrFreq;       // T9 record (tracking.py:255-275), stored by wave 2
                s_out[it & 1][1] = carrError;
                s_out[it & 1][2] = carrNco;
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
                s_out[it & 1][4] = (double)(pos_after + K.file_off);
                s_out[it & 1][5] = codeFreq;
                s_out[it & 1][6] = codeError;
                s_out[it & 1][7] = codeNco;
            }
            if (more) prep_code(K, codeFreq, rem_next, pos_after, s_st, s_blk, lane == 0);
        }
        else if (wave == 2 && member == 0) {
            // record (T9): the six sums of this block, and the scalar series of the previous block
            // (staged in LDS by the filter waves, visible since the last barrier) - off the critical path
            if (lane < 6) {
                const int series = (lane == 0) ? 4 : (lane == 1) ? 6 : (lane == 2) ? 3 : (lane == 3) ? 7 : (lane == 4) ? 5 : 8;
                o[series * m + it] = s_tot[lane];      // s_tot order: I_E Q_E I_P Q_P I_L Q_L
            }
            if (it > 0 && lane >= 8 && lane < 16 && lane != 11) {
                const int k = lane - 8;
                const int series = (k == 0) ? 2 : (k == 1) ? 11 : (k == 2) ? 12 : (k == 4) ? 0 : (k == 5) ? 1 : (k == 6) ? 9 : 10;
                o[series * m + (it - 1)] = s_out[(it - 1) & 1][k];
            }
        }
        done = it + 1;
        PROBE(11);  // loop filter (this wave's side) + next block's parameters
        __syncthreads();   // next block's parameters visible
        PROBE(12);  // barrier 3 (waits for the other filter wave)
        if (prof && tid == 0) {
            const long long tk4 = (long long)__builtin_amdgcn_s_memtime();
            pf_map += tk1 - tk0;
            pf_wait += tk2 - tk1;
            pf_red += tk3 - tk2;
            pf_flt += tk4 - tk3;
        }
    }
    if (tid == 0 && prof && member == 0) {
        prof[ch * 64 + 0] = pf_map;
        prof[ch * 64 + 1] = pf_wait;
        prof[ch * 64 + 2] = pf_red;
        prof[ch * 64 + 3] = pf_flt;
    }
    if (wave == 2 && member == 0 && done > 0 && lane >= 8 && lane < 16 && lane != 11) {
        const int k = lane - 8;
        const int series = (k == 0) ? 2 : (k == 1) ? 11 : (k == 2) ? 12 : (k == 4) ? 0 : (k == 5) ? 1 : (k == 6) ? 9 : 10;
        o[series * (long long)K.ms + (done - 1)] = s_out[(done - 1) & 1][k];
    }
#ifdef TRK_FINEPROF
    if (tid == 0 && prof && member == 0 && ch == 0) {
        for (int k = 0; k < 13; ++k) printf("[fineprof] probe %2d: %8.1f cycles/block\n", k, (double)fp[k] / K.ms);
    }
#endif
    if (tid == 0 && member == 0) ms_done[ch] = done;
}

// sgx_trk_spec.hip
void sgx_trk_spec_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                         double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                         int* err);

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
    {
        // cooperating workgroups per channel: one 256-thread workgroup per CU, all of a launch must be
        // resident at once (they wait for each other), so split * n_ch <= CU count
        int cus = 0;
        SGX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        // units needed by the longest possible block (samplesPerCode + 2 samples, worst alignment)
        K.n_units = (int)((c->n_code + 2 + 15 + 15) / 16 + TRK_THREADS - 1) / TRK_THREADS;
        int split = cus / (n_ch > 0 ? ((n_ch + 7) / 8) * 8 : 8);
        if (split > TRK_MAX_SPLIT) split = TRK_MAX_SPLIT;
        if (K.n_units > 16) {
            sgx_set_error("samplesPerCode %lld needs %d units, the tracking kernel holds 16", (long long)c->n_code,
                          K.n_units);
            return SGX_E_ARG;
        }
        if (split > K.n_units) split = K.n_units;
        if (split < 1) split = 1;
        const char* se = getenv("SGX_TRK_SPLIT");
        if (se && atoi(se) >= 1 && atoi(se) <= split) split = atoi(se);
        K.split = split;
        const char* fe = getenv("SGX_TRK_FASTX");
        K.fast_xcd = (fe && fe[0] == '0') ? 0 : 1;
        K.nb_base = (int)c->n_code - 3;
        for (int k = 0; k < 8; ++k) K.inv_nb[k] = 1.0 / (double)(K.nb_base + k);
        K.inv_fs = 1.0 / S.samplingFreq;
        K.inv_pi = 1.0 / M_PI;
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
    // device-side call state lives in one cached allocation: [channels | done | exchange | err | profile]
    const size_t sz_ch = ((sizeof(TrkChan) * (size_t)n_ch + 255) / 256) * 256;
    const size_t sz_done = ((sizeof(int) * (size_t)n_ch + 255) / 256) * 256;
    const size_t xch_bytes = sizeof(unsigned long long) * (size_t)n_ch * (2 * TRK_MAX_SPLIT * 12 + 16);
    const size_t sz_xch = ((xch_bytes + 255) / 256) * 256;
    const size_t sz_prof = sizeof(long long) * 64 * (size_t)n_ch;
    const size_t need = sz_ch + sz_done + sz_xch + 256 + sz_prof;
    if (c->trk_aux_cap < need) {
        if (c->d_trk_aux) hipFree(c->d_trk_aux);
        c->d_trk_aux = nullptr;
        c->trk_aux_cap = 0;
        if (hipMalloc(&c->d_trk_aux, need) != hipSuccess) {
            sgx_set_error("hipMalloc of %zu tracking state bytes failed", need);
            return SGX_E_NOMEM;
        }
        c->trk_aux_cap = need;
    }
    char* aux = (char*)c->d_trk_aux;
    TrkChan* d_ch = (TrkChan*)aux;
    int* d_done = (int*)(aux + sz_ch);
    unsigned long long* d_xch = (unsigned long long*)(aux + sz_ch + sz_done);
    int* d_err = (int*)(aux + sz_ch + sz_done + sz_xch);
    SGX_HIP(hipMemcpyAsync(d_ch, hc.data(), sizeof(TrkChan) * (size_t)n_ch, hipMemcpyHostToDevice, st));
    SGX_HIP(hipMemsetAsync(aux + sz_ch, 0, sz_done + sz_xch + 256, st));   // done, every polled word, err
    trk_fill_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, st>>>(c->d_trk_out, ms, (long long)elems);
    const char* pe = getenv("SGX_TRK_PROFILE");
    const bool want_prof = pe && pe[0] == '1';
    long long* d_prof = want_prof ? (long long*)(aux + sz_ch + sz_done + sz_xch + 256) : nullptr;
    const int n_blocks = ((n_ch + 7) / 8) * 8 * K.split;
    // SGX_TRK_SPEC=1 selects the experimental speculative pipeline (sgx_trk_spec.hip; needs exactly one unit
    // per member).  It reproduces the cooperative kernel's results but measured slower.
    const char* sp = getenv("SGX_TRK_SPEC");
    const bool use_spec = (K.split > 1 && K.split == K.n_units) && (sp && sp[0] == '1');
    if (want_prof) SGX_HIP(hipMemsetAsync(d_prof, 0, sizeof(long long) * 64 * (size_t)n_ch, st));
    hipEventRecord(c->ev[3], st);
    if (use_spec)
        sgx_trk_spec_launch(n_blocks, st, r->d, c->d_codes, d_ch, c->d_trk_out, d_done, K, d_prof, d_xch, d_err);
    else
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
        std::vector<long long> hp(64 * (size_t)n_ch);
        hipMemcpy(hp.data(), d_prof, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost);
        if (use_spec) {
            for (int i = 0; i < n_ch && i < 1; ++i) {
                const long long* q = &hp[64 * i];
                const double d = (double)ms;
                fprintf(stderr, "[sgx trk profile] ch %d spec: exact %lld/%lld wave-blocks\n", i, q[63], 4ll * ms);
                for (int mm = 0; mm < 2; ++mm) {
                    const long long* c2 = q + 8 * mm;
                    fprintf(stderr, "   CAR member %d: bookkeeping %.0f | wait-partials %.0f | rowsum+publish %.0f | poll %.0f | totals %.0f | PLL %.0f | tables+stores %.0f  (sum %.0f)\n",
                            mm * 5, c2[0] / d, c2[1] / d, c2[4] / d, c2[5] / d, c2[2] / d, c2[6] / d, c2[3] / d,
                            (c2[0] + c2[1] + c2[2] + c2[3] + c2[4] + c2[5] + c2[6]) / d);
                }
                fprintf(stderr, "   COD: wait-totals %.0f | DLL+params %.0f | prediction+stores %.0f\n", q[16] / d, q[18] / d, q[17] / d);
                fprintf(stderr, "   MAP w2: wait-params %.0f finalize %.0f fold %.0f wait-pred %.0f shadow %.0f | w4: wait-params %.0f finalize %.0f fold %.0f wait-pred %.0f shadow %.0f\n",
                        q[24] / d, q[25] / d, q[26] / d, q[27] / d, q[28] / d, q[32] / d, q[33] / d, q[34] / d, q[35] / d, q[36] / d);
            }
        } else
        for (int i = 0; i < n_ch && i < 4; ++i)
            fprintf(stderr, "[sgx trk profile] ch %d cycles/block: map %.0f wait %.0f reduce %.0f filter %.0f\n", i,
                    (double)hp[64 * i] / ms, (double)hp[64 * i + 1] / ms, (double)hp[64 * i + 2] / ms,
                    (double)hp[64 * i + 3] / ms);
    }
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
