// TrackingResult.track on gfx950, latency-mode kernel (reference tracking.py:13-295).
// A channel's block is cut into units of 256 groups x 16 samples and the workgroups of a channel (its MEMBERS)
// cooperate on it.  The per-block dependency chain
//     sums -> discriminators -> NCOs -> next block's parameters -> sums
// is built around what each link really depends on and what each instruction on it costs (a lone wave issues one
// fp64 / integer / DPP instruction every ~5.4 cycles whether or not it depends on the previous one, so the chain is
// shortened by REMOVING INSTRUCTIONS from it and by moving work to other waves and other CUs):
//
//   members   ARMS = 1 (round 3, used when 3 x channels x units workgroups fit the CUs): a unit has THREE workgroups,
//             one per correlator arm (early, prompt, late), on three CUs - a map lane then follows one code ramp,
//             applies that arm's chips to the samples as sign flips (two FMAs per sample) and reduces two sums.
//             ARMS = 3: one workgroup per unit follows all three ramps (the round-2 map; used when the CUs are scarce).
//   roles     a workgroup has 7 waves: 4 MAP waves (256 lanes, one 16-sample group each), a PLL wave, a DLL wave and a
//             RECORD wave.  One workgroup barrier per block hands the next block's parameters to the map waves.
//   map       the 16 bytes of a lane are loaded and converted one block ahead (the next block's first sample is known
//             when the current block starts).  The chip index at the group's first sample and the switch sample follow
//             from ONE fused evaluation t0 = ilo*step + start and the distance to the next chip boundary in samples,
//             u = (ceil(t0) - t0) / step, with step = codeFreq/fs: the reference's ramp t(i) = fl(fl(i*stp)+start)
//             (stp = linspace's delta/blk) differs from that real ramp by < 1e-12 chips, so ceil(t(i)) equals the real
//             ramp's for every sample of the group unless a boundary lies within that distance of a sample - excluded
//             when frac(u) is outside [1e-7, 1 - 1e-7] samples (2.7e-9 chips).  A wave in which any lane fails the test
//             (probability ~1e-5 per block; all of block 0, whose prompt ramp starts ON a boundary) takes the exact
//             search (ramp_setup) with the exact linspace steps, which the DLL wave posts right after the barrier.
//             Chip indices are therefore still bit-identical to code[int64(ceil(linspace(...)))] (tracking.py:166-188).
//             ARMS = 1: the chips a lane can meet are an 8-bit window of the packed sign table held in a register.
//   reduce    ARMS = 1: the lane's two sums go to fixed point (raw mantissa bits of fma(a, 2^28, 1.5 2^52): the bias
//             cancels in the low 48 bits), a transposing DPP reduction with fused 64-bit integer adds leaves the row
//             sums of I in the even and of Q in the odd lanes, and two lanes per row add them into one LDS word per sum
//             whose high bits count the arrivals: the lane that sees 15 earlier arrivals holds the member's total and
//             publishes it.  Integer adds are order-free, so the total is bit-identical whatever the arrival order.
//             ARMS = 3: six fp64 partials, DPP row reduction, LDS slots, the wave that arrives last folds and publishes.
//   exchange  one 8-byte granule per (sum, unit): {16-bit epoch tag | 48-bit two's-complement fixed point} written by
//             ONE aligned store, double-buffered by block parity, never reset.  The PLL and DLL waves of EVERY member
//             poll their part (L1-bypassing loads), add the payloads as integers and run their half of the loop filter
//             redundantly - bit-identical in all members, nothing to broadcast.
//   filter    only what the map waves need is on the chain.  DLL wave: discriminator, NCO, codeFreq/fs to 3 ulp, its
//             reciprocal, and the block length by a guarded division-free ceil (the IEEE division decides when the
//             quotient is within 6 ulp of an integer); the exact T1/T3/T4 arithmetic of the block (linspace steps, code
//             phase and first sample of the next block) follows AFTER the barrier, off the chain.  PLL wave: a
//             short-chain atan, NCO, and the carrier tables of the next block by ROTATION: the tables are computed in
//             full (fp64 turns reduction + sincos) for the OLD rate while the wave waits for the sums, and the rate
//             step dw turns each entry by exp(j dw m / fs), a degree-13 Taylor pair valid while |dw| m / fs <= 0.34 rad
//             (~50 Hz of NCO step; beyond it the full evaluation runs on the chain).
//   record    the 13 series values of a block are staged in LDS by member 0's filter waves and stored (to pinned host
//             memory, directly) by the record wave one block later, so no wave on the chain ever waits for a store.
//   abort     a poll that runs out of budget raises the channel's abort word; every member sees it in its next poll,
//             posts stop = 2 for the next block and the whole channel leaves the loop within one budget (the host then
//             repeats the launch with one workgroup per channel).
#include "sgx_trk_common.h"
#include "sgx_trk_math.h"

#include "sgx_trk2_parts.h"

// ================================ MAP (waves 0-3), all three arms in one workgroup ================================
// Member m of a channel owns units m, m + P, m + 2 P, ... (one unit when the CUs allow a workgroup per unit; several
// when there are more channels than that - up to all of them with one workgroup per channel).  The first unit's bytes
// are loaded and converted one block ahead; further units are loaded and converted when their turn comes.
template <int SB>
__device__ __forceinline__ int t2_map3_role(T2Shared& S, const int8_t* __restrict__ rec, long long rec_alloc, int ms,
                                            long long pos0, int member, int P, int n_units, bool uns, int tid,
                                            unsigned long long* __restrict__ xbase, bool fast, bool prof_on, bool prof_any,
                                            double fscale) {
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long long limit = rec_alloc - 16;                  // bytes: the last 16-byte word that may be loaded
    const int g = (tid & 255) + member * T2_MAP;             // the lane's group (first unit) inside the block's aligned window
    const long long lane_off = (long long)g * 16;
    T2_FP_DECL
    (void)prof_on;
    // state prepared one block ahead: the lane's 16 samples (first unit) as fp64 and where they sit in the block
    double xd[16];
    int i0, ilo;
    double ilod;
    T2Raw<SB> raw;
    int blk_pred;                    // block length the prepared samples are cut for

    // samples from block sample index END on belong to the next block: zeroed by an integer mask on the high dword
    // (integer samples: the low dword of their fp64 value is zero anyway; float samples: both dwords)
#define T2_CUT(XD, I0, END)                                                                                    \
    do {                                                                                                       \
        if ((I0) < (END) && (I0) + 16 > (END)) {                                                               \
            const int e_ = (END) - (I0);                                                                       \
            _Pragma("unroll") for (int b_ = 0; b_ < 16; ++b_)                                                  \
                (XD)[b_] = __hiloint2double(__double2hiint((XD)[b_]) & ((b_ - e_) >> 31),                      \
                                            SB >= 4 ? (__double2loint((XD)[b_]) & ((b_ - e_) >> 31)) : 0);    \
        }                                                                                                      \
    } while (0)
    // The block's last, partial group is cut in the shadow too, for the length the block will most likely have (the
    // current one): on the chain the one lane that holds it costs its whole member ~200 cycles every block, and every
    // member waits for that member.
#define T2_PREPARE(POS_NEXT, BLK_PRED)                                                                         \
    do {                                                                                                       \
        const int head_ = (int)((POS_NEXT) & 15);                                                              \
        i0 = g * 16 - head_;                                                                                   \
        ilo = i0 < 0 ? 0 : i0;                                                                                 \
        ilod = (double)ilo;                                                                                    \
        t2_convert<SB>(raw, i0, uns, xd, fscale);                                                                   \
        blk_pred = (BLK_PRED);                                                                                 \
        T2_CUT(xd, i0, blk_pred);                                                                              \
    } while (0)

    raw = t2_load<SB>(rec, (pos0 & ~15ll) + lane_off, limit);
    T2_PREPARE(pos0, S.code[0].blk);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T2Code& C = S.code[par];
        // one batch of LDS reads: chain part, early part, the lane's carrier phasors
        const int4 hd = *reinterpret_cast<const int4*>(&C.blk);      // blk, stop, inv_step
        const double step = C.step;
        const double inv_step = __hiloint2double(hd.w, hd.z);
        const double startE = C.start[0], startP = C.start[1], startL = C.start[2];
        const long long pos = C.pos;
        const T2Carr& CR = S.carr[par];
        const double2 w1 = CR.T[T2_W1 + (tid & 15)], w2 = CR.T[T2_W2 + ((tid >> 4) & 15)];
        if (hd.y) break;
        T2_FP_TOP
        __builtin_amdgcn_s_setprio(2);
        // the block's 16 sample phasors, once for all of the member's units: requested now, waited for in front of the
        // first accumulation (as in the arm-split map below)
        t2_v2d Bt[16];
        {
            const unsigned ba = (unsigned)(unsigned long long)&CR.T[T2_B];
#define T2_BLD(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(Bt[i]) : "v"(ba))
            T2_BLD(0); T2_BLD(1); T2_BLD(2); T2_BLD(3); T2_BLD(4); T2_BLD(5); T2_BLD(6); T2_BLD(7);
            T2_BLD(8); T2_BLD(9); T2_BLD(10); T2_BLD(11); T2_BLD(12); T2_BLD(13); T2_BLD(14); T2_BLD(15);
#undef T2_BLD
        }
        bool bt_landed = false;
        const int blk = hd.x;
        const long long pos_next = pos + blk;
        const T2Raw<SB> nraw = t2_load<SB>(rec, (pos_next & ~15ll) + lane_off, limit);   // next block's bytes (first unit)
        T2PROBE(prof_on, 0);   // parameters read, next block's load issued
        // W1[tid & 15] * W2[(tid >> 4) & 15]: the lane's phasor inside a unit
        const double lc = __builtin_fma(w1.x, w2.x, -(w1.y * w2.y));
        const double ls = __builtin_fma(w1.x, w2.y, w1.y * w2.x);
        if (__builtin_expect(blk != blk_pred, 0)) {
            // the block is a sample longer or shorter than predicted (about one block in ten): the lanes around its
            // end convert their bytes again and cut them at the real length
            const int lo_ = blk < blk_pred ? blk : blk_pred, hi_ = blk < blk_pred ? blk_pred : blk;
            if (__any(i0 < hi_ && i0 + 16 > lo_)) {
                t2_convert<SB>(raw, i0, uns, xd, fscale);
                T2_CUT(xd, i0, blk);
            }
        }
        double aIE = 0.0, aQE = 0.0, aIP = 0.0, aQP = 0.0, aIL = 0.0, aQL = 0.0;
        // ---- one 16-sample group of one unit: adds its six sums ----
        auto group = [&](const double (&x)[16], int gi0, int gilo, double gilod, double2 w3) {
            int kE, kP, kL, swE, swP, swL;
            bool bad = false;
            ramp_locate(startE, step, inv_step, gilod, gilo, kE, swE, bad);
            ramp_locate(startP, step, inv_step, gilod, gilo, kP, swP, bad);
            ramp_locate(startL, step, inv_step, gilod, gilo, kL, swL, bad);
            if (__builtin_expect(__any(bad && gi0 < blk), 0)) {
                // a chip boundary within 1e-7 samples of a sample somewhere in this wave: exact search with the exact
                // linspace steps (posted by the DLL wave right after the barrier)
                int budget = 1 << 20;
                while (lds_peek(&C.xflag) != it + 1 && --budget) __builtin_amdgcn_s_sleep(1);
                const double stpE = C.stp[0], stpP = C.stp[1], stpL = C.stp[2];
                ramp_setup(startE, stpE, inv_step, gilo, kE, swE);
                ramp_setup(startP, stpP, inv_step, gilo, kP, swP);
                ramp_setup(startL, stpL, inv_step, gilo, kL, swL);
            }
            // chips k1 and k1 + 1 of every ramp (two sign bits each); they are needed only after the accumulation
            const unsigned bE = chip_bits2(S.cbits, kE), bP = chip_bits2(S.cbits, kP), bL = chip_bits2(S.cbits, kL);
            // group-start phasor G = W1 * W2 * W3; a group that lies entirely beyond the block (its samples belong to
            // the next one) gets a zero phasor, i.e. adds nothing
            double gc = __builtin_fma(lc, w3.x, -(ls * w3.y));
            double gs = __builtin_fma(lc, w3.y, ls * w3.x);
            {
                const bool beyond = gi0 >= blk;
                gc = beyond ? 0.0 : gc;
                gs = beyond ? 0.0 : gs;
            }
            T2PROBE(prof_on, 1);   // switch samples resolved
            const double cE1 = __hiloint2double((int)(0x3FF00000u | (bE << 31)), 0), cE2 = __hiloint2double((int)(0x3FF00000u | ((bE >> 1) << 31)), 0);
            const double cP1 = __hiloint2double((int)(0x3FF00000u | (bP << 31)), 0), cP2 = __hiloint2double((int)(0x3FF00000u | ((bP >> 1) << 31)), 0);
            const double cL1 = __hiloint2double((int)(0x3FF00000u | (bL << 31)), 0), cL2 = __hiloint2double((int)(0x3FF00000u | ((bL >> 1) << 31)), 0);
            const int iend = gi0 + 16;
            int swmin = swE < swP ? swE : swP;
            swmin = swL < swmin ? swL : swmin;
            const bool eS = (swE == swmin), pS = (swP == swmin), lS = (swL == swmin);
            const bool odd = (swE < iend && !eS) || (swP < iend && !pS) || (swL < iend && !lS);
            if (!bt_landed) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bt[0]), "+v"(Bt[1]), "+v"(Bt[2]), "+v"(Bt[3]), "+v"(Bt[4]), "+v"(Bt[5]),
                             "+v"(Bt[6]), "+v"(Bt[7]), "+v"(Bt[8]), "+v"(Bt[9]), "+v"(Bt[10]), "+v"(Bt[11]), "+v"(Bt[12]), "+v"(Bt[13]),
                             "+v"(Bt[14]), "+v"(Bt[15]));
                bt_landed = true;
            }
            if (__builtin_expect(__any(odd), 0)) {
                // exact per-sample path (a ramp switches at a second position inside the group)
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int i = gi0 + b;
                    const double2 Bb = make_double2(Bt[b].x, Bt[b].y);
                    const double c = __builtin_fma(gc, Bb.x, -(gs * Bb.y));
                    const double s_ = __builtin_fma(gs, Bb.x, gc * Bb.y);
                    const double xs = s_ * x[b], xc = c * x[b];
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
                // samples b >= bsw come after the switch.  The samples are small integers, so their fp64 low dword is
                // zero and masking the HIGH dword alone zeroes one
                int bsw = swmin - gi0;
                bsw = bsw > 16 ? 16 : bsw;
                double Ac = 0.0, As = 0.0, Tc = 0.0, Ts = 0.0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const double2 Bb = make_double2(Bt[b].x, Bt[b].y);
                    Ac = __builtin_fma(x[b], Bb.x, Ac);
                    As = __builtin_fma(x[b], Bb.y, As);
                    const int keep = ~((b - bsw) >> 31);                  // all ones iff b >= bsw
                    const double xt = __hiloint2double(__double2hiint(x[b]) & keep, SB >= 4 ? (__double2loint(x[b]) & keep) : 0);
                    Tc = __builtin_fma(xt, Bb.x, Tc);
                    Ts = __builtin_fma(xt, Bb.y, Ts);
                }
                T2PROBE(prof_on, 2);   // 16-sample accumulation
                // rotate by the group phasor: cos part -> Q, sin part -> I (tracking.py:205-207)
                const double allQ = __builtin_fma(gc, Ac, -(gs * As));
                const double allI = __builtin_fma(gs, Ac, gc * As);
                const double tlQ = __builtin_fma(gc, Tc, -(gs * Ts));
                const double tlI = __builtin_fma(gs, Tc, gc * Ts);
                const double dE = eS ? (cE2 - cE1) : 0.0;
                const double dP = pS ? (cP2 - cP1) : 0.0;
                const double dL = lS ? (cL2 - cL1) : 0.0;
                aIE += __builtin_fma(dE, tlI, cE1 * allI);
                aQE += __builtin_fma(dE, tlQ, cE1 * allQ);
                aIP += __builtin_fma(dP, tlI, cP1 * allI);
                aQP += __builtin_fma(dP, tlQ, cP1 * allQ);
                aIL += __builtin_fma(dL, tlI, cL1 * allI);
                aQL += __builtin_fma(dL, tlQ, cL1 * allQ);
            }
        };
        group(xd, i0, ilo, ilod, CR.T[T2_W3]);
        // further units of this member (more channels than CUs for one workgroup per unit): loaded and converted now
#pragma unroll 1
        for (int j = 1; member + j * P < n_units; ++j) {
            const int gj = g + j * P * T2_MAP;
            const int j0 = gj * 16 - (int)(pos & 15);
            if (j0 >= blk) break;                              // (uniform: the whole unit lies beyond the block)
            const T2Raw<SB> rj = t2_load<SB>(rec, (pos & ~15ll) + (long long)gj * 16, limit);
            double xj[16];
            t2_convert<SB>(rj, j0, uns, xj, fscale);
            T2_CUT(xj, j0, blk);
            group(xj, j0, j0, (double)j0, CR.T[T2_W3 + j]);   // (j0 > 0: only the block's very first group starts before it)
        }
        T2PROBE(prof_on, 3);   // group finalisation
        // ---- transposing reduction inside each row of 16 lanes; exchange order I_P Q_P I_E Q_E I_L Q_L ----
        const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
        // xor 1: pairs (I_P, Q_P), (I_E, Q_E), (I_L, Q_L) -> a lane keeps the member selected by its bit 0
        double p = (b0 ? aQP : aIP) + dpp_mov<0xB1>(b0 ? aIP : aQP);
        double e = (b0 ? aQE : aIE) + dpp_mov<0xB1>(b0 ? aIE : aQE);
        double l = (b0 ? aQL : aIL) + dpp_mov<0xB1>(b0 ? aIL : aQL);
        // xor 2: pair (p, e) -> bit 1 selects; l is reduced plainly
        double pe = (b1 ? e : p) + dpp_mov<0x4E>(b1 ? p : e);
        l = l + dpp_mov<0x4E>(l);
        // rotations by 4 and 8 inside the row keep the low two lane bits: sums over the four lanes that share them
        pe = pe + dpp_mov<0x124>(pe);   // row_ror:4
        l = l + dpp_mov<0x124>(l);
        pe = pe + dpp_mov<0x128>(pe);   // row_ror:8
        l = l + dpp_mov<0x128>(l);
        // lanes 0..3 of every row: pe = row sum of word lane & 3 (I_P Q_P I_E Q_E); lanes 0..1: l = word 4 + (lane & 1).
        // Row sums go to LDS slots (plain stores, fixed slots: the member's total is formed in a fixed order).
        {
            const int r = lane & 15;
            double* slot = S.part[par][wave * 4 + (lane >> 4)];
            if (r < 4) slot[r] = pe;
            if (r < 2) slot[4 + r] = l;
        }
        T2PROBE(prof_on, 4);   // row reduction, slots written
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(&S.ticket[par][0], 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if ((ticket & 3u) == 3u) {
            // last of the four map waves: all 16 row sums of every word are in LDS.  Lane j adds slots 2p, 2p+1 of word
            // j >> 3 (p = j & 7), three DPP steps add the eight lanes of a word, lane 8 w converts the member's sum to
            // fixed point and publishes it as one granule.
            const int word = lane >> 3, pp = lane & 7;
            double v = 0.0;
            if (word < 6) v = S.part[par][2 * pp][word] + S.part[par][2 * pp + 1][word];
            v = v + dpp_mov<0xB1>(v);
            v = v + dpp_mov<0x4E>(v);
            v = v + dpp_mov<0x141>(v);   // row_half_mirror: the other quad of the eight
            if (pp == 0 && word < 6) {
                // {16-bit epoch tag | 48-bit two's-complement fixed point}, ONE aligned 8-byte store.  A member that
                // owns several units can hold more than 2^19 (2^28): the scale drops by the number of units, rounded up
                // to a power of two, and the reader's scale with it (t2_fix_of).
                const double t = __builtin_fma(v, t2_fix_of<SB>(P, n_units, uns), T2_MAGIC);
                const unsigned long long q = (unsigned long long)(__double_as_longlong(t) - __double_as_longlong(T2_MAGIC));
                const unsigned long long gran = ((unsigned long long)((unsigned)(it + 1) & 0xFFFFu) << 48) | (q & 0xFFFFFFFFFFFFull);
                granule_store(xbase + T2_XG + (par * 6 + word) * T2_XLINE + member, gran, fast);
            }
            if (prof_any && lane == 0) S.tpub[par] = (long long)__builtin_amdgcn_s_memtime();
        }
        T2STAMP(prof_on, 5);   // published (or handed to the wave that publishes)
        // ---- shadow: prepare the next block with this block's rates ----
        __builtin_amdgcn_s_setprio(0);
        raw = nraw;
        T2_PREPARE(pos_next, blk);
        T2STAMP(prof_on, 6);   // next block prepared
        __builtin_amdgcn_sched_barrier(0);   // (the conversions above stay in the barrier's shadow)
        wg_barrier();
        __builtin_amdgcn_sched_barrier(0);
        T2STAMP(prof_on, 7);   // waiting for the loop filter
    }
    T2_FP_PRINT(prof_on && lane == 0, 0, 8)
    return it;
#undef T2_CUT
#undef T2_PREPARE
}

// ================================ MAP (waves 0-3), one workgroup per (unit, arm) ================================
// The lane follows ONE code ramp.  Its chips enter as sign flips of the samples: with c1 the chip at the group's first
// sample and c2 the next one, sum_b c(b) x_b B_b = c1 * sum_b f_b x_b B_b, f_b = +1 before the switch sample and
// c1 c2 after it; c1 goes onto the group phasor.  Two FMAs, one compare and one select per sample.
template <int SB>
__device__ __forceinline__ int t2_map1_role(T2Shared& S, const int8_t* __restrict__ rec, long long rec_alloc, int ms,
                                            long long pos0, int unit, int arm, int tid,
                                            unsigned long long* __restrict__ xbase, bool fast, double step_nom, double spacing,
                                            bool uns, bool prof_on, bool prof_any) {
    const int lane = tid & 63;
    const long long limit = rec_alloc - 16;                  // bytes: the last 16-byte word that may be loaded
    const int g = (tid & 255) + unit * T2_MAP;               // the lane's group inside the block's aligned window
    const long long lane_off = (long long)g * 16;
    const int wbase = (arm == 1) ? 0 : ((arm == 0) ? 2 : 4); // exchange order I_P Q_P I_E Q_E I_L Q_L
    T2_FP_DECL
    (void)prof_on;
    // The chips this lane can meet: the group's first sample sits at t0 = (16 g - head) step + rem + offset chips with
    // head in [0, 15] and rem in [0, step), i.e. inside a 0.46-chip interval while the code NCO stays near its basis.
    // Eight sign bits from chip `ws` on cover a code-rate error of 0.4 % (4 kHz); beyond that the wave reads LDS.
    int ws;
    unsigned win;
    {
        const double off = (arm == 0) ? -spacing : ((arm == 2) ? spacing : 0.0);
        const int kb = (int)ceil((double)(16 * g - 15) * step_nom + off - 0.02);
        ws = kb - 1 < -1 ? -1 : kb - 1;
        const int bi = ws + 1;                               // bit k + 1 of the packed table is chip k
        const unsigned lo = S.cbits[bi >> 5], hi = S.cbits[(bi >> 5) + 1];
        win = (unsigned)((((unsigned long long)hi << 32) | lo) >> (bi & 31)) & 0xFFu;
    }
    double magic = T2_MAGIC;
    T2_PIN(magic);
    // state prepared one block ahead
    unsigned xh[16];                 // high dwords of the samples as fp64, zero outside the block
    int i0, ilo;
    double ilod;
    T2Raw<SB> raw;
    int blk_pred;                    // block length the prepared samples are cut for
    long long pos = pos0;            // first sample of the current block
    T2Ptr<SB> ptr_pred;              // where this lane's 16 samples of the NEXT block lie if that block is blk_pred long

#define T2_PREPARE1(BLK_PRED)                                                                                  \
    do {                                                                                                       \
        const int head_ = (int)(pos & 15);                                                                     \
        i0 = g * 16 - head_;                                                                                   \
        ilo = i0 < 0 ? 0 : i0;                                                                                 \
        ilod = (double)ilo;                                                                                    \
        blk_pred = (BLK_PRED);                                                                                 \
        t2_convert_hi<SB>(raw, i0, blk_pred, uns, xh);                                                              \
        ptr_pred = t2_ptr<SB>(((pos + blk_pred) & ~15ll) + lane_off, limit);                                  \
        T2_PIN(ilod);                                                                                          \
        asm volatile("" : "+v"(ptr_pred.a));                                                                   \
    } while (0)

    raw = t2_load<SB>(rec, (pos0 & ~15ll) + lane_off, limit);
    T2_PREPARE1(S.code[0].blk);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T2Code& C = S.code[par];
        // one batch of LDS reads: chain part, this arm's ramp start, the lane's carrier phasors
        const int4 hd = *reinterpret_cast<const int4*>(&C.blk);          // blk, stop, inv_step
        const double2 sp = *reinterpret_cast<const double2*>(&C.step);   // step, start_arm
        const T2Carr& CR = S.carr[par];
        const double2 w1 = CR.T[T2_W1 + (tid & 15)], w2 = CR.T[T2_W2 + ((tid >> 4) & 15)], w3 = CR.T[T2_W3];
        T2_USE(hd.z); T2_USE(sp.x); T2_USE(w1.x); T2_USE(w2.x); T2_USE(w3.x);   // one batch, one wait
        if (hd.y) break;
        T2_FP_TOP
        __builtin_amdgcn_s_setprio(2);
        // the block's 16 sample phasors: requested NOW (the compiler sinks these reads to their uses, four ahead of the
        // FMAs, with a wait in front of every pair), waited for once in front of the accumulation - they arrive while the
        // ramp is located: 52.5 -> 51.4 ms.  (asm: the reads and their one wait are invisible to the compiler's own
        // counting, which in-order LDS returns make safe; the registers are live from the read to the wait statement)
        t2_v2d Bt[16];
        {
            const unsigned ba = (unsigned)(unsigned long long)&CR.T[T2_B];
#define T2_BLD(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(Bt[i]) : "v"(ba))
            T2_BLD(0); T2_BLD(1); T2_BLD(2); T2_BLD(3); T2_BLD(4); T2_BLD(5); T2_BLD(6); T2_BLD(7);
            T2_BLD(8); T2_BLD(9); T2_BLD(10); T2_BLD(11); T2_BLD(12); T2_BLD(13); T2_BLD(14); T2_BLD(15);
#undef T2_BLD
        }
        const int blk = hd.x;
        const double inv_step = __hiloint2double(hd.w, hd.z);
        // next block's bytes: the address was prepared for the predicted length
        T2Ptr<SB> nptr = ptr_pred;
        if (__builtin_expect(blk != blk_pred, 0)) nptr = t2_ptr<SB>(((pos + blk) & ~15ll) + lane_off, limit);
        const T2Raw<SB> nraw = t2_load_at<SB>(rec, nptr);
        T2PROBE(prof_on, 0);   // parameters read, next block's load issued
        int k1, isw;
        bool bad = false;
        ramp_locate(sp.y, sp.x, inv_step, ilod, ilo, k1, isw, bad);
        int sh = k1 - ws;
        unsigned bits;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(i0 < blk && (bad || (unsigned)sh > 6u)) != 0, 0)) {
            // a chip boundary within 1e-7 samples of a sample somewhere in this wave (or a chip outside the window):
            // exact search with the exact linspace step (posted by the DLL wave right after the barrier), chips from LDS
            int budget = 1 << 20;
            while (lds_peek(&C.xflag) != it + 1 && --budget) __builtin_amdgcn_s_sleep(1);
            ramp_setup(sp.y, C.stp[arm], inv_step, ilo, k1, isw);
            bits = chip_bits2(S.cbits, k1);
        } else {
            bits = (win >> (sh & 7)) & 3u;
        }
        // switch sample inside the group; a group whose two chips are equal has none
        const unsigned flip = (bits ^ (bits >> 1)) & 1u;
        int bsw = isw - i0;
        bsw = flip ? bsw : 16;
        const unsigned s1 = bits << 31;                       // sign of the first chip (bit set: -1)
        if (__builtin_expect(blk != blk_pred, 0)) {
            // the block is a sample longer or shorter than predicted (about one block in ten): the lanes around its
            // end convert their bytes again and cut them at the real length
            const int lo_ = blk < blk_pred ? blk : blk_pred, hi_ = blk < blk_pred ? blk_pred : blk;
            if (__any(i0 < hi_ && i0 + 16 > lo_)) t2_convert_hi<SB>(raw, i0, blk, uns, xh);
        }
        // group-start phasor G = W1[tid & 15] * W2[(tid >> 4) & 15] * W3, times the first chip
        double gc, gs;
        {
            const double lc = __builtin_fma(w1.x, w2.x, -(w1.y * w2.y));
            const double ls = __builtin_fma(w1.x, w2.y, w1.y * w2.x);
            gc = __builtin_fma(lc, w3.x, -(ls * w3.y));
            gs = __builtin_fma(lc, w3.y, ls * w3.x);
            gc = __hiloint2double((int)((unsigned)__double2hiint(gc) ^ s1), __double2loint(gc));
            gs = __hiloint2double((int)((unsigned)__double2hiint(gs) ^ s1), __double2loint(gs));
        }
        T2PROBE(prof_on, 1);   // switch sample resolved, group phasor
        double Ac = 0.0, As = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bt[0]), "+v"(Bt[1]), "+v"(Bt[2]), "+v"(Bt[3]), "+v"(Bt[4]), "+v"(Bt[5]), "+v"(Bt[6]),
                     "+v"(Bt[7]), "+v"(Bt[8]), "+v"(Bt[9]), "+v"(Bt[10]), "+v"(Bt[11]), "+v"(Bt[12]), "+v"(Bt[13]), "+v"(Bt[14]),
                     "+v"(Bt[15]));
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const double2 Bb = make_double2(Bt[b].x, Bt[b].y);
            const unsigned h = (b >= bsw) ? (xh[b] ^ 0x80000000u) : xh[b];
            const double xs = __hiloint2double((int)h, 0);
            Ac = __builtin_fma(xs, Bb.x, Ac);
            As = __builtin_fma(xs, Bb.y, As);
        }
        T2PROBE(prof_on, 2);   // 16-sample accumulation
        // rotate by the group phasor: cos part -> Q, sin part -> I (tracking.py:205-207)
        const double aQ = __builtin_fma(gc, Ac, -(gs * As));
        const double aI = __builtin_fma(gs, Ac, gc * As);
        // fixed point: the raw bits of fma(a, 2^28, 1.5 2^52) are bias + round(a 2^28); sums of them carry the sum of
        // the integers in their low 48 bits whatever the bias adds up to (sixteen biases leave the low 52 bits alone).
        // Two-byte samples: 2^24 in the lanes (a member's total needs 52 bits), rounded to the granule's 2^19 once.
        const double lane_fix = (SB == 1) ? (uns ? T2_FIX * 0.5 : T2_FIX) : T2_FIX16 * 32.0;
        constexpr unsigned long long res_mask = (SB == 1) ? 0xFFFFFFFFFFFFull : 0xFFFFFFFFFFFFFull;
        // (three-address FMAs with the bias in a register pair: the accumulating form wants it copied in front of each)
        double tI, tQ;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(tI) : "v"(aI), "s"(lane_fix), "v"(magic));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(tQ) : "v"(aQ), "s"(lane_fix), "v"(magic));
        const unsigned long long qI = (unsigned long long)__double_as_longlong(tI);
        const unsigned long long qQ = (unsigned long long)__double_as_longlong(tQ);
        T2PROBE(prof_on, 3);   // group finalisation
        // transposing reduction inside each row of 16 lanes: even lanes end with the row's I, odd lanes with its Q
        const bool b0 = (lane & 1) != 0;
        unsigned long long v = dpp_addl_xor1(b0 ? qI : qQ, b0 ? qQ : qI);
        v = dpp_addl_xor2(v, v);
        v = dpp_addl_ror4(v, v);
        v = dpp_addl_ror8(v, v);
        T2PROBE(prof_on, 4);   // row reduction
        if ((lane & 15) < 2) {
            const unsigned long long mine = (v & res_mask) | (1ull << 56);
            const unsigned long long prev = atomicAdd(&S.acc[par][lane & 1], mine);
            if ((prev >> 56) == 15ull) {
                // the sixteenth arrival (4 waves x 4 rows): this lane holds the member's total
                unsigned long long tot = prev + mine;
                if constexpr (SB == 2) tot = (unsigned long long)((((long long)(tot << 12) >> 12) + 16) >> 5);
                S.acc[par][lane & 1] = 0ull;
                const unsigned long long gran = ((unsigned long long)((unsigned)(it + 1) & 0xFFFFu) << 48) | (tot & 0xFFFFFFFFFFFFull);
                granule_store(xbase + T2_XG + (par * 6 + wbase + (lane & 1)) * T2_XLINE + unit, gran, fast);
                if (prof_any && (lane & 1) == 0) S.tpub[par] = (long long)__builtin_amdgcn_s_memtime();
            }
        }
        T2STAMP(prof_on, 5);   // published (or handed to the lane that publishes)
        // ---- shadow: prepare the next block with this block's length ----
        __builtin_amdgcn_s_setprio(0);
        raw = nraw;
        pos += blk;
        T2_PREPARE1(blk);
        T2STAMP(prof_on, 6);   // next block prepared
        __builtin_amdgcn_sched_barrier(0);   // (the conversions above stay in the barrier's shadow)
        wg_barrier();
        __builtin_amdgcn_sched_barrier(0);
        T2STAMP(prof_on, 7);   // waiting for the loop filter
    }
    T2_FP_PRINT(prof_on && lane == 0, 0, 8)
    return it;
#undef T2_PREPARE1
}

// ================================ PLL (wave 4) ================================
template <int SB>
__device__ __forceinline__ int t2_pll_role(T2Shared& S, const TrkConst& K, const TrkChan& cc, int unit, int mslot, bool owner,
                                           int lane, int P, int ch, unsigned long long* __restrict__ xbase,
                                           int* __restrict__ err, bool prof_on, long long* __restrict__ prof) {
    // tracking.py:123-130
    long long acc_map = 0, acc_xch = 0, acc_flt = 0, t_top = 0, t_arr = 0;   // SGX_TRK_PROFILE=1: per-member phase times
    double carrBasis = cc.acquiredFreq;
    double remCarr = 0.0, w_cur = (cc.acquiredFreq * 2.0) * M_PI, oldCarrNco = 0.0, oldCarrErr = 0.0;
    const double two_pi = 2 * M_PI;
    // constants of the call in registers (kernel arguments would be re-fetched through the scalar cache on the chain)
    double k_a = K.k_carr_a, k_b = K.k_carr_b, inv_2pi = K.inv_2pi, c_hi = K.inv_2pifs_hi, c_lo = K.inv_2pifs_lo,
           inv_fs = K.inv_fs, fs = K.fs;
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(inv_2pi); T2_PIN(c_hi); T2_PIN(c_lo); T2_PIN(inv_fs); T2_PIN(carrBasis);
    T2_PIN(fs);
    // the largest rate step the rotation takes: the farthest table entry is sample n_units * UNIT of the block
    double dw_max = SGX_ROT_MAX / ((double)(K.n_units * TRK_UNIT) * K.inv_fs);
    T2_PIN(dw_max);
    const int ms = K.ms;
    // polynomial coefficients in registers (a constant the compiler materialises in front of every use is an
    // instruction on the chain)
    SgxAtanCoef ak = sgx_atan_coef();
    SgxRotCoef rk = sgx_rot_coef();
    T2_PIN(ak.c0); T2_PIN(ak.c1); T2_PIN(ak.c2); T2_PIN(ak.c3); T2_PIN(ak.c4); T2_PIN(ak.c5); T2_PIN(ak.c6); T2_PIN(ak.c7); T2_PIN(ak.c8);
    T2_PIN(rk.s0); T2_PIN(rk.s1); T2_PIN(rk.s2); T2_PIN(rk.s3); T2_PIN(rk.s4); T2_PIN(rk.s5);
    T2_PIN(rk.c0); T2_PIN(rk.c1); T2_PIN(rk.c2); T2_PIN(rk.c3); T2_PIN(rk.c4); T2_PIN(rk.c5);
    // lane 0 of a row adds the bits of 1.5 2^52 to its payload: the row's integer sum then IS the double 1.5 2^52 + sum
    const int bias_hi = ((lane & 15) == 0) ? 0x43380000 : 0;
    double unfix = 1.0 / t2_fix_of<SB>(P, K.n_units, K.uns != 0);
    T2_PIN(unfix);
    const bool w3 = lane >= 48;
    // lane = 16 word + unit polls that unit's granule of I_P (word 0, row 0) / Q_P (word 1, row 1)
    const bool mine = (lane < 32) && ((lane & 15) < P);
    unsigned long long* const xabort = xbase + T2_XABORT;
    // record values of the block just finished (member 0), posted after the barrier: carrFreq I_P Q_P pllDiscr pllDiscrFilt
    double r_cf = 0.0, r_ip = 0.0, r_qp = 0.0, r_err = 0.0, r_nco = 0.0;
    T2_FP_DECL
    (void)prof_on;
    __builtin_amdgcn_s_setprio(3);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T2Code& C = S.code[par];
        const int4 hd = *reinterpret_cast<const int4*>(&C.blk);
        const long long pos = C.pos;
        if (owner && it > 0) {
            if (lane == 0) {
                double* R = S.rec[par ^ 1];      // T9 record (tracking.py:255-275) of block it - 1, stored by the record wave
                R[2] = r_cf;
                R[3] = r_ip;
                R[7] = r_qp;
                R[11] = r_err;
                R[12] = r_nco;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) lds_poke(&S.rflag[0], it);
        }
        if (hd.y) break;
        T2_FP_TOP
        if (prof) t_top = (long long)__builtin_amdgcn_s_memtime();
        // ---- before the sums arrive ----
        // carrier phase at the end of this block (T5), exact remainder by FMA
        const int blk = hd.x;
        const int head_next = (int)((pos + blk) & 15);
        double rc;
        {
            const double arg_end = w_cur * div_rn((double)blk, fs, inv_fs) + remCarr;   // blk / fs, correctly rounded
            const double kq = floor(arg_end * inv_2pi);
            rc = __builtin_fma(-kq, two_pi, arg_end);
            if (rc < 0.0) rc += two_pi;
            if (rc >= two_pi) rc -= two_pi;
        }
        // the next block's table entry of this lane at the CURRENT rate, in full; the sums then only turn it
        const int mi = t2_carr_mult(lane, unit, P, head_next);
        double mf = (double)mi * inv_fs;             // m / fs: the rate step dw turns the entry by dw * mf radians
        double cs_p, sn_p;
        t2_carr_entry(c_hi, c_lo, inv_2pi, w_cur, rc, mi, w3, cs_p, sn_p);
        T2_PIN(cs_p); T2_PIN(sn_p); T2_PIN(mf); T2_PIN(rc);   // (keeps all of this ahead of the wait)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long* gp = xbase + T2_XG + (par * 6 + ((lane >> 4) & 1)) * T2_XLINE + (lane & 15);
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
#ifdef T2_POLL2
        {   // two polls in flight, half a round trip apart
            unsigned long long xa_ = 0, xb_ = 0;
            if (mine) xa_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (mine) xb_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (xa_ >> 48) == tag)) { x = xa_; break; }
                if (mine) xa_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (xb_ >> 48) == tag)) { x = xb_; break; }
                if ((--budget & 31) == 0) {
                    if (budget == 0 || lds_peek(&S.flag[1]) != 0) {
                        gave_up = true;
                        break;
                    }
                }
            }
        }
#else
        for (;;) {
            if (mine) x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(!mine || (x >> 48) == tag)) break;
            if ((--budget & 31) == 0) {
                if (budget == 0 || lds_peek(&S.flag[1]) != 0) {
                    gave_up = true;
                    break;
                }
            }
        }
#endif
        T2STAMP(prof_on, 8);   // waiting for the sums
        if (prof) {
            t_arr = (long long)__builtin_amdgcn_s_memtime();
            const long long tp = lds_peek64(&S.tpub[par]);
            acc_map += tp - t_top;       // barrier release -> this member's publish
            acc_xch += t_arr - tp;       // this member's publish -> every member's sums visible
        }
        // sum of the units' payloads (integers: exact, order-free), rows of 16 lanes; lanes that poll nothing hold 0
        (void)bias_hi;
        const double v = t2_sum48_row(x) * unfix;
        const double I_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                            __builtin_amdgcn_readlane(__double2loint(v), 0));
        const double Q_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16),
                                            __builtin_amdgcn_readlane(__double2loint(v), 16));
        // T7 PLL (tracking.py:223-235); atan(Q/I) / 2 / pi as one multiplication by RN(1 / (2 pi)) (1.5 ulp)
        const double carrError = sgx_atan_ratio_k(Q_P, I_P, ak) * inv_2pi;
        const double carrNco = oldCarrNco + k_a * (carrError - oldCarrErr) + carrError * k_b;
        const double carrFreq = carrBasis + carrNco;
        const double w_new = (carrFreq * 2.0) * M_PI;
        oldCarrNco = carrNco;
        oldCarrErr = carrError;
        T2PROBE(prof_on, 9);   // discriminator + NCO
        // carrier tables of the next block: the prepared entry turned by the rate step (exact: w_new - w_cur is)
        if (it + 1 < ms) {
            const double dw = w_new - w_cur;
            double cs, sn;
            if (__builtin_expect(fabs(dw) <= dw_max, 1)) {
                double es, ec;
                sgx_rot_small(dw * mf, rk, es, ec);
                cs = __builtin_fma(cs_p, ec, -(sn_p * es));
                sn = __builtin_fma(cs_p, es, sn_p * ec);
            } else {
                t2_carr_entry(c_hi, c_lo, inv_2pi, w_new, rc, mi, w3, cs, sn);
            }
            // B, W1, W2 are consecutive 16-entry tables, the W3 of this member's units follow
            S.carr[par ^ 1].T[lane] = make_double2(cs, sn);
        }
        w_cur = w_new;
        remCarr = rc;
        r_cf = carrFreq;
        r_ip = I_P;
        r_qp = Q_P;
        r_err = carrError;
        r_nco = carrNco;
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            atomicCAS(err, 0, 1 + ch);
            __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        T2STAMP(prof_on, 10);  // carrier tables
        wg_barrier();
        __builtin_amdgcn_s_setprio(0);   // what follows until the next wait is off the chain: let the map waves issue first
        if (prof) acc_flt += (long long)__builtin_amdgcn_s_memtime() - t_arr;   // sums visible -> barrier released (both filter waves done)
        T2STAMP(prof_on, 11);
    }
    if (owner && it > 0 && it == ms) {
        // (when the loop ran out of blocks, the last block's record values are still in registers)
        if (lane == 0) {
            double* R = S.rec[(it - 1) & 1];
            R[2] = r_cf;
            R[3] = r_ip;
            R[7] = r_qp;
            R[11] = r_err;
            R[12] = r_nco;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) lds_poke(&S.rflag[0], it);
    }
    if (prof && lane == 0) {
        prof[ch * T2_PROF_STRIDE + mslot] = acc_map;
        prof[ch * T2_PROF_STRIDE + 64 + mslot] = acc_xch;
        prof[ch * T2_PROF_STRIDE + 128 + mslot] = acc_flt;
    }
    T2_FP_PRINT(prof_on && lane == 0, 8, 12)
    return it;
}

// ================================ DLL (wave 5) ================================
// Lanes work in parallel on the three ramps: lane & 3 = 0 early, 1 prompt, 2 late (3 repeats prompt); uniform results
// come from lane 1 / lane 0.
struct T2DllConst {
    double fs, inv_fs, code_len, spacing;
    double inv_nb_lane;     // RN(1 / (nb_base + (lane & 7))): reciprocals of the plausible block lengths, one per lane
    int nb_base;
    long long rec_len;
};

template <int SB, int ARMS>
__device__ __forceinline__ int t2_dll_role(T2Shared& S, const TrkConst& K, const T2DllConst& D, long long pos0, int blk0,
                                           int stop0, int arm, bool owner, int lane, int P, int ch,
                                           unsigned long long* __restrict__ xbase, int* __restrict__ err, bool prof_on,
                                           long long file_off) {
    // tracking.py:114-121; block 0's chain part and ramp starts were posted before the loop
    double oldCodeNco = 0.0, oldCodeErr = 0.0;
    double k_a = K.k_code_a, k_b = K.k_code_b, basis = K.code_basis;
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(basis);
    const int ms = K.ms;
    // the block being processed: code phase at its start, first sample, code frequency, length
    double rem = 0.0, cf = K.code_basis;
    long long pos = pos0;
    int blk = blk0, stop = stop0;
    const int l4 = lane & 3;
    const double off = (l4 == 0) ? -D.spacing : ((l4 == 2) ? D.spacing : 0.0);   // rem - spc == rem + (-spc) exactly
    const int lim3 = K.n_units * TRK_UNIT - 15;            // the longest block the units of the launch hold
    const int bias_hi = ((lane & 15) == 0) ? 0x43380000 : 0;   // (see the PLL wave)
    double unfix = 1.0 / t2_fix_of<SB>(P, K.n_units, K.uns != 0);
    T2_PIN(unfix);
    // lane = 16 row + unit polls that unit's granule of word 2 + row: rows I_E, Q_E, I_L, Q_L
    const bool mine = (lane & 15) < P;
    unsigned long long* const xabort = xbase + T2_XABORT;
    // record values of the block just finished (member 0), posted after the barrier
    double r_v = 0.0, r_cf = 0.0, r_err = 0.0, r_nco = 0.0;
    T2_FP_DECL
    (void)prof_on;
    __builtin_amdgcn_s_setprio(3);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        T2Code& C = S.code[par];
        T2Code& N = S.code[par ^ 1];
        if (owner && it > 0) {
            double* R = S.rec[par ^ 1];
            // rows hold I_E | Q_E | I_L | Q_L -> series 4, 6, 5, 8 (order of _native.SERIES)
            if ((lane & 15) == 0) R[lane == 0 ? 4 : (lane == 16 ? 6 : (lane == 32 ? 5 : 8))] = r_v;
            if (lane == 0) {
                R[0] = (double)(pos * SB + file_off);   // position after block it - 1 = first sample of block it
                R[1] = r_cf;
                R[9] = r_err;
                R[10] = r_nco;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) lds_poke(&S.rflag[1], it);
        }
        if (stop) break;
        T2_FP_TOP
        // ---- before the sums arrive: the exact arithmetic of this block (T1, T3), which only the exact search reads
        const double step = div_rn(cf, D.fs, D.inv_fs);                             // codeFreq / fs
        const double nb = (double)blk;
        const double span = nb * step;                                              // blksize * codePhaseStep
        const int ki = blk - D.nb_base;
        const bool known = (ki >= 0 && ki < 8);
        const int kq = __builtin_amdgcn_readfirstlane(ki) & 7;
        const double ynb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(D.inv_nb_lane), kq),
                                            __builtin_amdgcn_readlane(__double2loint(D.inv_nb_lane), kq));
        // np.linspace(start, stop, blk, endpoint=False): delta = stop - start; step = delta / blk
        const double start = rem + off;
        const double d = ((span + rem) + off) - start;
        double stp;
        if (__builtin_expect(known, 1)) stp = div_rn(d, nb, ynb);
        else stp = d / nb;
        if (lane < 3) C.stp[lane] = stp;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) lds_poke(&C.xflag, it + 1);
        // code phase and first sample of the next block (T4) and that block's ramp starts
        const double t_last = ramp_at(blk - 1, stp, start);
        const double rn_lane = (t_last + step) - 1023.0;                            // meaningful in the prompt lane
        const double rem_next = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rn_lane), 1),
                                                 __builtin_amdgcn_readlane(__double2loint(rn_lane), 1));
        const long long pos_next = pos + blk;
        if (lane < 3) N.start[lane] = rem_next + off;
        if (ARMS == 1 && lane == arm) N.start_arm = rem_next + off;
        if (lane == 0) N.pos = pos_next;
        double a_next = D.code_len - rem_next;                                      // (1023 - rem) of T1
        // the longest next block the record (stop 1 beyond it) and the units of the launch (stop 3) hold: ONE compare on
        // the chain, which of the two it was is sorted out in the (rare) branch
        const long long room = D.rec_len - pos_next;
        const int lim1 = room > (long long)0x3FFFFFFF ? 0x3FFFFFFF : (room < 0 ? 0 : (int)room);
        unsigned lim = (unsigned)(lim1 < lim3 ? lim1 : lim3);
        T2_PIN(a_next); T2_PIN(lim);
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long* gp = xbase + T2_XG + (par * 6 + 2 + (lane >> 4)) * T2_XLINE + (lane & 15);
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x = 0, xa = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
#ifdef T2_POLL2
        {
            unsigned long long xa_ = 0, xb_ = 0;
            if (mine) xa_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (mine) xb_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (xa_ >> 48) == tag)) { x = xa_; break; }
                if (mine) xa_ = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (xb_ >> 48) == tag)) { x = xb_; break; }
                if ((--budget & 15) == 0) {
                    xa = __hip_atomic_load(xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (xa != 0 || budget == 0) {
                        gave_up = true;
                        break;
                    }
                }
            }
        }
#else
        for (;;) {
            if (mine) x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(!mine || (x >> 48) == tag)) break;
            if ((--budget & 15) == 0) {
                xa = __hip_atomic_load(xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (xa != 0 || budget == 0) {
                    gave_up = true;
                    break;
                }
            }
        }
#endif
        T2STAMP(prof_on, 12);  // waiting for the sums
        // T8 DLL (tracking.py:238-251).  Row r of the wave holds the units' payloads of I_E | Q_E | I_L | Q_L: integer
        // row sums (exact, order-free; lanes that poll nothing hold 0), every lane of a row then has its row's total.
        // The discriminator runs on the rows as they are: no lane shuffles besides two row broadcasts.
        (void)bias_hi;
        const double v = t2_sum48_row(x) * unfix;   // I_E | Q_E | I_L | Q_L by row
        const double sq = v * v;
        const double e2 = sq + dpp_bcast<0x142, 0xA>(sq);    // rows 1, 3: I_E^2 + Q_E^2, I_L^2 + Q_L^2
        const double mag = sgx_sqrt1_pos(e2);                // rows 1, 3: E, L (two zero envelopes: NaN, as in the reference)
        const double oth = dpp_bcast<0x143, 0xC>(mag);       // rows 2, 3: E
        const double ce_lane = sgx_div1(oth - mag, oth + mag);       // row 3: (E - L) / (E + L)
        const double codeError = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ce_lane), 48),
                                                  __builtin_amdgcn_readlane(__double2loint(ce_lane), 48));
        const double codeNco = oldCodeNco + k_a * (codeError - oldCodeErr) + codeError * k_b;
        const double cf_new = basis - codeNco;
        oldCodeNco = codeNco;
        oldCodeErr = codeError;
        T2PROBE(prof_on, 13);  // discriminator + NCO
        // chain part of the next block: its length, the ramps' slope and the slope's reciprocal
        double step_a, inv_step;
        const int blk_n = sgx_block_length(a_next, cf_new, D.fs, D.inv_fs, step_a, inv_step);
        int stop_n = 0;
        if (__builtin_expect((unsigned)(blk_n - 1) >= lim || gave_up, 0)) {      // (blk <= 0 wraps to a huge number)
            stop_n = gave_up ? 2 : ((blk_n <= 0 || blk_n > lim1) ? 1 : 3);
            if (lane == 0 && stop_n == 3) {
                atomicOr(err, TRK_ERR_RANGE);
                atomicCAS(err + 1, 0, 1 + ch);
            }
        }
        if (lane == 0) {
            *reinterpret_cast<int4*>(&N.blk) = make_int4(blk_n, stop_n, __double2loint(inv_step), __double2hiint(inv_step));
            N.step = step_a;
        }
        rem = rem_next;
        pos = pos_next;
        cf = cf_new;
        blk = blk_n;
        stop = stop_n;
        r_v = v;
        r_cf = cf_new;
        r_err = codeError;
        r_nco = codeNco;
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            if (xa == 0) {
                atomicCAS(err, 0, 1 + ch);
                __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        T2STAMP(prof_on, 14);  // next block's code parameters
        wg_barrier();
        __builtin_amdgcn_s_setprio(0);
        T2STAMP(prof_on, 15);
    }
    if (owner && it > 0 && it == ms) {
        double* R = S.rec[(it - 1) & 1];
        if ((lane & 15) == 0) R[lane == 0 ? 4 : (lane == 16 ? 6 : (lane == 32 ? 5 : 8))] = r_v;
        if (lane == 0) {
            R[0] = (double)(pos * SB + file_off);
            R[1] = r_cf;
            R[9] = r_err;
            R[10] = r_nco;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) lds_poke(&S.rflag[1], it);
    }
    T2_FP_PRINT(prof_on && lane == 0, 12, 16)
    return it;
}

// ================================ RECORD (wave 6) ================================
// Stores block k's 13 series values once both filter waves have posted them (rflag >= k + 1): one block behind.
__device__ __forceinline__ void t2_rec_store(T2Shared& S, int k, long long m, int lane, double* __restrict__ o,
                                             int* __restrict__ err, int ch) {
    int budget = 1 << 16;
    while ((lds_peek(&S.rflag[0]) < k + 1 || lds_peek(&S.rflag[1]) < k + 1) && --budget) __builtin_amdgcn_s_sleep(2);
    if (budget == 0) {   // the filter waves never posted block k: flag the channel (the host repeats or reports), no stale row
        if (lane == 0) atomicCAS(err, 0, 1 + ch);
        return;
    }
    if (lane < SGX_NUM_SERIES) o[lane * m + k] = S.rec[k & 1][lane];
}

// A record that is still streaming in from a file (K.mark: the device-side watermark, bytes resident so far, advanced in
// copy-stream order): while block `it` is processed, everything block it + 1 can touch - its own samples and the
// prefetch of the block after it - must be resident.  This wave, idle otherwise, checks that one block ahead, before
// it arrives at the barrier that starts block it + 1.  Copies advance in 32 MiB steps (a multiple of every cache-line
// size), so no line is ever fetched half written; the watermark is cached in a register, so the uncached load is
// issued once per ~900 blocks.
template <int SB>
__device__ __forceinline__ int t2_rec_role(T2Shared& S, const TrkConst& K, int pad, int ch, bool owner, int lane,
                                           double* __restrict__ o, int* __restrict__ err, unsigned long long mark_seen) {
    const long long m = K.ms;
    const int ms = K.ms;
    const long long span = (2ll * K.n_units * TRK_UNIT + 64) * SB;   // bytes: a block and the window of the prefetch behind it
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        if (S.code[par].stop) break;
        if (owner && it > 0) t2_rec_store(S, it - 1, m, lane, o, err, ch);
        {   // (a resident record: mark_seen is all ones and this returns at once)
            const long long need = S.code[par].pos * SB + pad + 3 * span;
            wait_mark(K.mark, need < K.rec_len ? need : K.rec_len, mark_seen, err, ch);
        }
        wg_barrier();
    }
    return it;
}

// ARMS = 1: a channel has 3 P members, member m = arm * P + unit; ARMS = 3: P members, member = unit.
template <int SB, int ARMS>
__global__ __launch_bounds__(T2_THREADS) void trk2_kernel(const int8_t* __restrict__ rec, const int8_t* __restrict__ codes,
                                                          const TrkChan* __restrict__ chans, double* __restrict__ out,
                                                          int* __restrict__ ms_done, TrkConst K,
                                                          long long* __restrict__ prof,
                                                          unsigned long long* __restrict__ xch, int* __restrict__ err) {
    __shared__ T2Shared S;
    const int P = K.split;
    const int PM = (ARMS == 1) ? 3 * P : P;
    const int bq = blockIdx.x >> 3, br = blockIdx.x & 7;
    const int ch = br + 8 * (bq / PM);
    const int member = bq % PM;
    const int unit = member % P;
    const int arm = (ARMS == 1) ? member / P : 1;
    const bool owner = member == 0;            // the member that records the channel's series
    if (ch >= K.n_ch) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const TrkChan cc = chans[ch];
    if (cc.prn == 0) {
        if (tid == 0 && owner) ms_done[ch] = 0;
        return;
    }
    unsigned long long* __restrict__ xbase = xch + (long long)ch * T2_XCH_STRIDE;   // granules, abort word, placement granules
    unsigned long long* const xabort = xbase + T2_XABORT;
    const bool prof_on = (owner && ch == 0 && prof != nullptr && (wave == 0 || wave == 4 || wave == 5));

    // ---- placement: are all members of the channel on one XCD (one L2)?  Then the exchange may stay in that L2.
    if (tid < 4) {
        S.flag[tid] = 0;
        S.rflag[tid] = 0;
        S.ticket[tid >> 1][tid & 1] = 0;
        S.acc[tid >> 1][tid & 1] = 0ull;
    }
    if (tid < 2) {
        S.code[tid].xflag = 0;
        S.code[tid].start_arm = 0.0;
    }
    __syncthreads();
    if (wave == 4) {
        unsigned long long* pl = xbase + T2_XPLACE;
        const unsigned me = xcc_id();
        if (lane == 0) __hip_atomic_store(pl + member, 0xC0DE000000000000ull | me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long x = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
        for (;;) {
            if (lane < PM) x = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane >= PM || (x >> 48) == 0xC0DE;
            if (__all(ok)) break;
            if (--budget == 0) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const bool same = __all(lane >= PM || (unsigned)(x & 0xF) == me);
        if (lane == 0) {
            S.flag[0] = (same && !gave_up && K.fast_xcd != 0) ? 1 : 0;
            if (gave_up) {   // a member is not resident: give the channel up at once (the host repeats with split 1)
                S.flag[1] = 1;
                atomicCAS(err, 0, 1 + ch);
                __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    for (int i = tid; i < 1032; i += T2_THREADS) {
        const int k = i - 1;                       // extended-code index; chip = code[(k - 1) mod 1023]
        const int j = (k - 1 + 2 * 1023) % 1023;
        S.chip[i] = (codes[(cc.prn - 1) * 1023 + j] > 0) ? 0x3FF00000u : 0xBFF00000u;
    }
    __syncthreads();
    if (tid < 40) {
        unsigned w = 0;
        for (int bb = 0; bb < 32; ++bb) {
            const int i = tid * 32 + bb;
            if (i < 1032 && (S.chip[i] >> 31)) w |= 1u << bb;
        }
        S.cbits[tid] = w;
    }
    __syncthreads();
    const bool fast = S.flag[0] != 0;
    const bool dead = S.flag[1] != 0;

    // block 0 parameters (tracking.py:114-130): chain part and ramp starts
    T2DllConst D;
    int blk0 = 0, stop0 = 0;
    if (wave == 5) {
        D.fs = K.fs;
        D.inv_fs = K.inv_fs;
        D.code_len = K.code_len;
        D.spacing = K.spacing;
        D.inv_nb_lane = 1.0 / (double)(K.nb_base + (lane & 7));
        D.nb_base = K.nb_base;
        D.rec_len = (K.rec_len - cc.pad) / SB;      // samples on the channel's grid (cc.pad: its byte shift, SB = 2 only)
        double step_a, inv_step;
        blk0 = sgx_block_length(K.code_len - 0.0, K.code_basis, D.fs, D.inv_fs, step_a, inv_step);
        const int lim3 = K.n_units * TRK_UNIT - 15;
        stop0 = dead ? 2 : ((blk0 <= 0 || cc.pos0 + blk0 > D.rec_len) ? 1 : ((blk0 > lim3) ? 3 : 0));
        const double off = ((lane & 3) == 0) ? -K.spacing : (((lane & 3) == 2) ? K.spacing : 0.0);
        if (lane < 3) S.code[0].start[lane] = 0.0 + off;
        if (ARMS == 1 && lane == arm) S.code[0].start_arm = 0.0 + off;
        if (lane == 0) {
            *reinterpret_cast<int4*>(&S.code[0].blk) = make_int4(blk0, stop0, __double2loint(inv_step), __double2hiint(inv_step));
            S.code[0].step = step_a;
            S.code[0].pos = cc.pos0;
            if (blk0 > lim3) {
                atomicOr(err, TRK_ERR_RANGE);
                atomicCAS(err + 1, 0, 1 + ch);
            }
        }
    }
    if (wave == 4) {
        double cs, sn;
        t2_carr_entry(K.inv_2pifs_hi, K.inv_2pifs_lo, K.inv_2pi, (cc.acquiredFreq * 2.0) * M_PI, 0.0,
                      t2_carr_mult(lane, unit, P, (int)(cc.pos0 & 15)), lane >= 48, cs, sn);
        S.carr[0].T[lane] = make_double2(cs, sn);
    }
    // a streaming record: block 0 and the prefetch of block 1 must be resident before the first loads
    unsigned long long mark_seen = K.mark ? 0ull : ~0ull;
    if (wave == 6 && K.mark) {
        const long long need = cc.pos0 * SB + cc.pad + 3 * (2ll * K.n_units * TRK_UNIT + 64) * SB;
        wait_mark(K.mark, need < K.rec_len ? need : K.rec_len, mark_seen, err, ch);
    }
    __syncthreads();

    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    int done;
    if (wave < 4) {
        if constexpr (ARMS == 1)
            done = t2_map1_role<SB>(S, rec + cc.pad, K.rec_alloc, K.ms, cc.pos0, unit, arm, tid, xbase, fast,
                                    K.code_basis / K.fs, K.spacing, K.uns != 0, prof_on, prof != nullptr);
        else
            done = t2_map3_role<SB>(S, rec + cc.pad, K.rec_alloc, K.ms, cc.pos0, unit, P, K.n_units, K.uns != 0, tid, xbase, fast, prof_on, prof != nullptr, K.fscale);
    } else if (wave == 4)
        done = t2_pll_role<SB>(S, K, cc, unit, member, owner, lane, P, ch, xbase, err, prof_on, prof);
    else if (wave == 5)
        done = t2_dll_role<SB, ARMS>(S, K, D, cc.pos0, blk0, stop0, arm, owner, lane, P, ch, xbase, err, prof_on, K.file_off + cc.pad);
    else
        done = t2_rec_role<SB>(S, K, cc.pad, ch, owner, lane, o, err, mark_seen);

    // a channel that was given up reports the blocks completed before the abort
    const bool aborted = S.code[done & 1].stop == 2;
    if (wave == 6 && owner && done > 0 && !aborted) t2_rec_store(S, done - 1, (long long)K.ms, lane, o, err, ch);
    if (aborted && done > 0) done -= 1;
    if (tid == 0 && owner) ms_done[ch] = done;
}

// arms = 1: 3 * split members per channel (n_blocks = 8-padded channels x 3 x split); arms = 3: split members.
// lds_pad: extra dynamic LDS per workgroup, so that a CU holds ONE workgroup of the launch (members must not share
// a CU: they would share its issue ports).
void sgx_trk2_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                     double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err,
                     int sample_bytes, int arms, int lds_pad) {
#define T2_LAUNCH(SBV, ARMSV)                                                                                          \
    do {                                                                                                               \
        if (lds_pad > 0)                                                                                               \
            (void)hipFuncSetAttribute((const void*)trk2_kernel<SBV, ARMSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      lds_pad);                                                                        \
        trk2_kernel<SBV, ARMSV><<<n_blocks, T2_THREADS, (size_t)(lds_pad > 0 ? lds_pad : 0), st>>>(rec, codes, chans, out, done, K, prof, xch, err); \
    } while (0)
    if (sample_bytes == 4) {          // float32 / float64: one workgroup per unit, all three arms (arms == 3)
        T2_LAUNCH(4, 3);
    } else if (sample_bytes == 8) {
        T2_LAUNCH(8, 3);
    } else if (sample_bytes == 2) {
        if (arms == 1) T2_LAUNCH(2, 1);
        else T2_LAUNCH(2, 3);
    } else {
        if (arms == 1) T2_LAUNCH(1, 1);
        else T2_LAUNCH(1, 3);
    }
#undef T2_LAUNCH
}
