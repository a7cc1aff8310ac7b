// TrackingResult.track on gfx950, the SPECULATIVE latency-mode kernel (reference tracking.py:13-295; SURVEY.md section 9
// T1-T9).  Rounds 4-5.  What the per-block chain  sums -> discriminators -> NCOs -> next block's parameters -> sums  of
// sgx_trk2.hip still held was the whole map: block k + 1's samples were multiplied and added only after block k's loop
// filter had posted the block's code rate, carrier rate and length.  But those three move by tiny, bounded amounts from
// block to block, and everything else about block k + 1 - its first sample, its code phase, its carrier start phase - is
// known as soon as block k's OWN parameters are.  So here block k + 1 is accumulated in the SHADOW of block k's exchange
// and loop filter with block k's rates, and the chain holds corrections only:
//
//   carrier   the lane's 16 sample phasors B_b were those of rate w_k; the true ones are B_b e^{j eps b}, eps = (w_{k+1} -
//             w_k) / fs.  The lane keeps the moments M0 = sum x_b B_b and M1 = sum x_b B_b (b - 7.5); with the group
//             phasor taken at the group's CENTRE (the PLL wave turns W3 by 7.5 samples more),
//             sum x_b B_b e^{j eps (b - 7.5)} = M0 + j eps M1 - eps^2/2 M2 + ...,  |eps| 7.5 <= 1e-4.  The second-order
//             term is 10.625 eps^2 M0 up to parts that cancel (the lock signal is constant over the group; the
//             double-frequency term has sum (-1)^b (b - 7.5)^2 = 0): a factor common to every lane, arm and unit, which
//             the discriminators (ratios) do not see and the RECORDED sums are multiplied by off the chain; third order
//             < 1e-13.  The group phasor G' = W1' W2' W3' comes from the tables the PLL wave ROTATES by the rate step.
//   code      at this sampling rate the chip boundaries of the three correlator arms, taken together, are half a chip
//             = 18.7 samples apart, so a 16-sample group meets at most ONE boundary of ANY arm (the host checks the
//             rate and the correlator spacing; otherwise sgx_trk2.hip runs).  One lane therefore serves all three arms
//             at the price of one: it accumulates the moments of ALL its samples and of those IN FRONT of the boundary,
//             and arm a's sum is alpha_a (all) + beta_a (front) with alpha, beta in {+-1, 0, +-2} from the arm's two chips.
//             A rate step moves the boundary by a fraction of a sample (|d codeFreq| < 27 Hz: less than one), so at most
//             one sample changes sides: the pass keeps the two candidates' terms and the final pass adds the one whose
//             threshold the new boundary position u' = (K - t(ilo - 1)) / step crossed.
//   fallback  a boundary within 1e-7 samples of a sample (1e-5 of the blocks; all of block 0), two arms switching at
//             different samples of one group: the wave takes the DIRECT path - per-sample chips from the exact linspace
//             ramps, this block's own sample phasors - on the chain.  A block whose length was mispredicted (~2 %; the
//             wave that holds its end), a rate step beyond the rotation's range (the PLL wave evaluates the tables in
//             full and says so): the wave accumulates again on the chain.  Chip indices are therefore bit-identical to
//             code[int64(ceil(linspace(...)))] (tracking.py:166-188) always.
//
// Members: one workgroup per unit of 128 groups x 16 samples (20 units at 38.192 Msps: the 8-channel launch occupies 160 CUs):
// two SETS of two MAP waves, a PLL wave, a DLL wave and a RECORD wave.  The sets take the blocks in turn.
//
// Round 5: the pass runs TWO blocks ahead, in two parts around the block barrier.  With the pass of block k + 1 in the
// shadow of block k (round 4) a set's pass - 380 instructions, 2 400 cycles alone and 3 200 next to the filter waves - filled
// the period: every cycle taken off the chain made the pass the critical path.  Now a set runs the final pass of block k,
// then part A of the pass of block k + 2 (edges, boundary, chips, candidates: what needs the block's geometry), arrives at
// the barrier, and runs part B (the sixteen samples' moments) in the first half of block k + 1's period: no map wave is
// near the chain any more, and the SIMDs are quiet while the loop filter runs.  Block k + 2's geometry is block k's moved on
// with block k's rates: its start is off by a sample in 3 % of the blocks and its length in 5 %.  The samples' ABSOLUTE
// positions and phases do not depend on where the block was taken to start, so the lanes keep their moments with their block
// indices moved; the one or two samples that enter or leave at the block's first and last lane are PATCHED in (raw byte, the
// pass's sample phasor); the rate step is the one over two blocks (eps = (w_{k+2} - w_k) / fs, W3 turned by 7.5 samples of
// it).  What the expansion leaves out grows with the square of that step: the front part carries its own second-order
// factor, and steps beyond 20 Hz (the pull-in) are evaluated exactly - tables in full, accumulate again - because the
// residue of those first blocks sat in the code NCO's integrator for the rest of the run (code phase 1e-11 chips from the
// reference's after 10 000 blocks: enough to put a sample 3e-12 chips from a chip boundary on the other side).
// Exchange: granules {16-bit epoch | 48-bit fixed point, 2^30}, order-free integer sums, redundant filters in every member;
// an arm's I and Q of a unit side by side; the filter waves keep THREE looks in flight (the PLL wave 8-byte loads, the DLL
// wave one 16-byte load per look; reserved registers v[244:255]).  Record path and abort protocol are those of
// sgx_trk2.hip.
#include "sgx_trk2_parts.h"

// issue priorities off the chain: the speculative pass (parts A and B) and the filter waves' work in front of their polls;
// on the chain: the final pass 2, the filter waves from their poll to the barrier 3
#ifndef T3_PRIO_SPEC
#define T3_PRIO_SPEC 0
#endif
#ifndef T3_PRIO_PRE
#define T3_PRIO_PRE 1
#endif
#define T3_STR2(x) #x
#define T3_STR(x) T3_STR2(x)
#define T3_LANES 128               // map lanes = groups per unit (two waves)
#define T3_UNIT (T3_LANES * 16)    // samples per unit
#define T3_THREADS 448             // 2 x 2 map waves (two SETS, alternating blocks) + PLL wave (4) + DLL wave (5) + record wave (6)
#define T3_MAXP 32                 // units per channel
#ifndef T3_XLINE
#define T3_XLINE 32                // granules per line: [2 parities][6 sums] lines of 32 units
#endif
#define T3_XABORT (12 * T3_XLINE)
#define T3_XPLACE (12 * T3_XLINE + 8)
#ifndef T3_XCH_STRIDE
#define T3_XCH_STRIDE 512          // words per channel (sgx_trk.hip sizes the allocation with the same figure)
#endif
// Fixed-point scale of a granule's 48-bit payload: 2^30 (sgx_trk2.hip: 2^28).  A lane's six sums are rounded to it before
// the order-free integer reductions, 2 387 lanes x 20 units of them per block: at 2^28 that rounding was the largest
// difference between this kernel's sums and the reference's own (9e-13 relative against 4e-13 from the reference's carrier
// argument; the code NCO disagrees by an ulp in 3 % of the blocks because of it, and the code phases of two
// implementations then drift apart until a sample within 1e-11 chips of a chip boundary falls on different sides).
// A unit's total must stay below 2^17 (the default scene's prompt sums: 7 500 per unit).  The guard (round 6; round 5's
// looked at the units' prompt payloads only, which have already WRAPPED when a total passes 1.5 x 2^17): any arm's sum over
// samples is at most the sum of their magnitudes - every sample enters once, with a chip of +-1 and a phasor of modulus 1 -
// so while the 2 048 bytes of a unit's window add up to less than 2^17 in magnitude (a mean of 64 of the 127 an int8 sample
// can reach: a record that clips all the time), no arm of the unit can reach 2^17.  A resident record is scanned ONCE by the
// host side (sgx_trk.hip: if_mag_bound, cached in the record's handle; a record beyond the bound goes to sgx_trk2.hip, said
// on stderr); a record that is still streaming in is watched by the RECORD wave, block by block (v_sad_u8 on two 16-byte
// loads per lane, a DPP reduction: it has the time; the map waves have none), and flags TRK_ERR_SCALE: the host repeats the
// launch with sgx_trk2.hip and says so.  Offset-binary records (uint8, read as they lie: a DC of 128 that the correlation
// cancels but a magnitude bound cannot) keep the round-5 guard: the units' prompt payloads beyond HALF the room, seen by the
// PLL wave behind the barrier.
#ifndef T3_FIX
#define T3_FIX 1073741824.0
#endif

struct __attribute__((aligned(128))) T3Code {   // code side of a block's parameters (DLL wave -> everybody), by block parity
    // chain part: written right before the barrier that starts the block
    int blk;
    int stop;               // 1: the record ends inside this block (tracking.py:159-163); 2: a member gave up waiting;
                            // 3: the block does not fit the units of the launch
    double inv_step;        // ~1 / step (2^-40): distances to chip boundaries in samples
    double step;            // codeFreq / fs to 3 ulp: the slope of the real ramps (the guards cover the difference)
    double pad0;
    // early part: posted right after the barrier that starts the PREVIOUS block
    double start[3];        // ramp starts E, P, L (exact; T3)
    long long pos;          // record index of the block's first sample
    // exact part: posted right after the barrier that starts the block (the direct path reads it)
    double stp[3];          // linspace steps E, P, L (tracking.py:166-188)
    int xflag;              // block number + 1 once stp[] is valid
    int pad1;
};

struct T3Carr {   // carrier side (PLL wave -> map waves), by block parity: (cos, sin)(2 pi r m), r = turns per sample
    double2 T[64];    // [0..15] B: m = b;  [16..31] W1: m = 16 a;  [32..39] W2: m = 256 r;  [48] W3: m = 2048 u - head + 7.5 d
    double eps;       // (w - w_prev) / fs: the block's moments were accumulated with the PREVIOUS block's B
    double respec;    // != 0: the rate step was beyond the rotation's range: tables in full, eps = 0, accumulate again
};

struct T3Shared {
    unsigned cbits[40];             // packed sign bits of the extended code: bit k + 1 set where chip k is -1
    T3Code code[2];
    T3Carr carr[2];
    unsigned long long acc[2][8];   // {arrival count << 56 | 48-bit fixed-point sum} of the six sums, by block parity
    uint4 scratch[256];             // a map lane's 16 bytes (outside the block: zeros), for reading single samples back by a dynamic index
    uint4 raw[256];                 // the same bytes as loaded: a block that starts or ends a sample or two from where the pass put it
    double2 btab[2][16];            // the sample phasors a set's pass ran with (the carr[] slot is another block's by the final pass)
    double rec[2][16];              // a block's 13 series values (member 0), stored one block later
    int flag[4];                    // [0] same-XCD placement, [1] abort seen by this workgroup
    int rflag[4];                   // [0] PLL wave, [1] DLL wave: number of blocks whose record values are in rec[]
    long long tpub[2];              // (profiling) time stamp of the member's publish, by block parity
#ifdef T3_TIMELINE
    unsigned long long tl[6][2][8]; // (diagnosis) sums of the waves' event stamps [wave][block parity][event]
#endif
};

// -DTRK_WAVEPROF: how long after the barrier's release each wave of (channel 0, member 0) arrives at the next one
#ifdef TRK_WAVEPROF
#ifndef T3_WB_UNIT
#define T3_WB_UNIT 0
#endif
#define T3_WB_ON(unit, ch) ((unit) == T3_WB_UNIT && (ch) == 0)
#define T3_WB_DECL long long wb_acc[2] = {0, 0}, wb_t = (long long)__builtin_amdgcn_s_memtime();
#define T3_WB(on)                                                                \
    do {                                                                         \
        if (on) {                                                                \
            const long long d_ = (long long)__builtin_amdgcn_s_memtime() - wb_t;  \
            wb_acc[0] += (it & 1) ? 0 : d_;   /* (constant indices: registers) */ \
            wb_acc[1] += (it & 1) ? d_ : 0;                                       \
        }                                                                        \
        wg_barrier();                                                            \
        if (on) wb_t = (long long)__builtin_amdgcn_s_memtime();                  \
    } while (0)
#define T3_WB_PRINT(on, name, n)                                                 \
    if ((on) && (threadIdx.x & 63) == 0) {                                       \
        unsigned hw_;                                                            \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));        \
        printf("[waveprof] %s wave %d (simd %u cu %u): release -> arrival %.1f cycles on even blocks, %.1f on odd blocks\n", name, (int)(threadIdx.x >> 6), (hw_ >> 4) & 3u, (hw_ >> 8) & 15u, 2.0 * (double)wb_acc[0] / (n), 2.0 * (double)wb_acc[1] / (n)); \
    }
#else
#define T3_WB_ON(unit, ch) false
#define T3_WB_DECL
#define T3_WB(on) wg_barrier()
#define T3_WB_PRINT(on, name, n)
#endif

// -DT3_TIMELINE: where a block's period goes.  The waves of channel 0 add s_memtime stamps of their events into LDS
// counters by block parity over the blocks >= T3_TL_FROM; the means are ABSOLUTE times (one counter per XCD, and a channel
// lives on one XCD), so that differences between waves and members are mean distances between their events.  A stamp
// waits for its counter value (~60 cycles on the chain), so a build takes only the events of its masks -DT3_TL_MAP=,
// -DT3_TL_PLL=, -DT3_TL_DLL= (bit e: event e) plus the PLL wave's release stamp, which every other time is relative to
// (a barrier releases all waves of the workgroup together).  tools/r6_timeline.sh builds and runs the set of variants.
#ifdef T3_TIMELINE
#ifndef T3_TL_FROM
#define T3_TL_FROM 2000
#endif
#ifndef T3_TL_MAP
#define T3_TL_MAP 0
#endif
#ifndef T3_TL_PLL
#define T3_TL_PLL 0
#endif
#ifndef T3_TL_DLL
#define T3_TL_DLL 0
#endif
#define T3_TL_ADD(wave_, e, v)                                                                                      \
    do {                                                                                                            \
        const unsigned long long v_ = (unsigned long long)(v);                                                      \
        if ((threadIdx.x & 63) == 0) atomicAdd(&S.tl[wave_][it & 1][e], v_);                                        \
    } while (0)
#define T3_TL(mask, on, wave_, e)                                                                                   \
    do {                                                                                                            \
        if ((((mask) >> (e)) & 1) && (on) && it >= T3_TL_FROM) T3_TL_ADD(wave_, e, __builtin_amdgcn_s_memtime());    \
    } while (0)
#define T3_TL_SET(mask, on, wave_, e, v)                                                                            \
    do {                                                                                                            \
        if ((((mask) >> (e)) & 1) && (on) && it >= T3_TL_FROM) T3_TL_ADD(wave_, e, v);                               \
    } while (0)
#else
#define T3_TL(mask, on, wave_, e) do { } while (0)
#define T3_TL_SET(mask, on, wave_, e, v) do { } while (0)
#endif

// Sum of the 48-bit payloads (two's complement) of the granules of lanes 0..31 / 32..63, as a double, in rows 1 / 3 of the
// wave: the payload is split into a low limb of 24 bits and a sign-extended high limb, whose sums over 32 lanes fit 32 bits -
// two INDEPENDENT chains of one DPP add per step instead of one chain of add + add-with-carry (the carry is a second
// dependent instruction in each of the five steps of the PLL wave's chain behind its poll); hi 2^24 + lo is exact.
__device__ __forceinline__ double t3_sum48_half(unsigned long long x) {
    const unsigned xl = (unsigned)x, xh = (unsigned)(x >> 32);
    unsigned lo = xl & 0xFFFFFFu;
    unsigned hi = (unsigned)((int)(__builtin_amdgcn_alignbit(xh, xl, 24) << 8) >> 8);   // bits 24..47, sign-extended
    unsigned a, b;
#define T3_S48(d0, d1, s0, s1, ctl)                                                               \
        "v_add_u32_dpp " d0 ", " s0 ", " s0 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32_dpp " d1 ", " s1 ", " s1 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "s_nop 0\n\t"
    asm volatile(
        "s_nop 1\n\t"
        T3_S48("%2", "%3", "%0", "%1", "quad_perm:[1,0,3,2]")
        T3_S48("%0", "%1", "%2", "%3", "quad_perm:[2,3,0,1]")
        T3_S48("%2", "%3", "%0", "%1", "row_half_mirror")
        T3_S48("%0", "%1", "%2", "%3", "row_mirror")
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
        : "+v"(lo), "+v"(hi), "=&v"(a), "=&v"(b));
#undef T3_S48
    return __builtin_fma((double)(int)hi, 16777216.0, (double)lo);
}

// the same for two granules per lane (the DLL wave: I in x1, Q in x2): four independent chains, no wait state to fill
__device__ __forceinline__ void t3_sum48_half2(unsigned long long x1, unsigned long long x2, double& v1, double& v2) {
    const unsigned x1l = (unsigned)x1, x1h = (unsigned)(x1 >> 32), x2l = (unsigned)x2, x2h = (unsigned)(x2 >> 32);
    unsigned l1 = x1l & 0xFFFFFFu, l2 = x2l & 0xFFFFFFu;
    unsigned h1 = (unsigned)((int)(__builtin_amdgcn_alignbit(x1h, x1l, 24) << 8) >> 8);
    unsigned h2 = (unsigned)((int)(__builtin_amdgcn_alignbit(x2h, x2l, 24) << 8) >> 8);
    unsigned a, b, c, d;
#define T3_S48(d0, d1, d2, d3, s0, s1, s2, s3, ctl)                                               \
        "v_add_u32_dpp " d0 ", " s0 ", " s0 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32_dpp " d1 ", " s1 ", " s1 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32_dpp " d2 ", " s2 ", " s2 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32_dpp " d3 ", " s3 ", " s3 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile(
        "s_nop 1\n\t"
        T3_S48("%4", "%5", "%6", "%7", "%0", "%1", "%2", "%3", "quad_perm:[1,0,3,2]")
        T3_S48("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "quad_perm:[2,3,0,1]")
        T3_S48("%4", "%5", "%6", "%7", "%0", "%1", "%2", "%3", "row_half_mirror")
        T3_S48("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "row_mirror")
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf"
        : "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d));
#undef T3_S48
    v1 = __builtin_fma((double)(int)h1, 16777216.0, (double)l1);
    v2 = __builtin_fma((double)(int)h2, 16777216.0, (double)l2);
}

__device__ __forceinline__ int t3_carr_mult(int lane, int unit, int head) {
    const int sel = lane >> 4, idx = lane & 15;
    return (sel == 3) ? (T3_UNIT * unit - head) : (idx << (4 * sel));
}

// ================================ MAP (waves 0-1: set 0, waves 2-3: set 1) ================================
// A lone wave pays ~9 cycles per operation for a compare + select pair through VCC and ~5 for anything else
// (tools/ubench_fp64.hip), so everything per-lane below is ARITHMETIC: masks from shifts and bit-field extracts, selects
// on the high dwords of doubles whose low dwords are zero (+-1, +-2, 0, half-integers) by v_bfi, clamps by min / max.
//
// One ramp for the three arms (the correlator spacing is half a chip; the host checks): in HALF chips, h(i) = 2 (i step +
// rem), the prompt ramp's boundaries are the even integers and the early and late ramps' (t -+ 1/2) the odd ones.  With
// kh = ceil(h(ilo - 1)) the next boundary of any arm, u = (kh - h(ilo - 1)) / (2 step) samples behind the anchor,
// kb = kh >> 1 and c0, c1, c2 the chips kb, kb + 1, kb + 2:
//     kh even (a prompt boundary):      E = c0 (all)                  P = c1 (all) + (c0 - c1)(front)   L = c1 (all)
//     kh odd  (early and late, shared): E = c1 (all) + (c0 - c1)(front)  P = c1 (all)                  L = c2 (all) + (c1 - c2)(front)
// (the arms' own ramp starts differ from rem -+ 1/2 by a rounding, 1e-16 chips; the guard is 1e-7 samples = 2.7e-9 chips).

// 3 x (I, Q) fixed-point values per lane -> row sums: after it lane r of a row of 16 holds the row's sum of word r
// (r < 4: I_P Q_P I_E Q_E) in `pe` and lanes 4, 5 (every lane, by its parity) those of I_L, Q_L in `l`.  One block, the
// steps of the three chains interleaved so that no DPP source was written by either of the two instructions before it.
__device__ __forceinline__ void t3_reduce6(const unsigned long long (&q)[6], unsigned long long odd1, unsigned long long odd2,
                                           unsigned long long& pe, unsigned long long& l) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, p0, p1, e0, e1, l0, l1, x0, x1, x2, x3;
    const unsigned q0l = (unsigned)q[0], q0h = (unsigned)(q[0] >> 32), q1l = (unsigned)q[1], q1h = (unsigned)(q[1] >> 32);
    const unsigned q2l = (unsigned)q[2], q2h = (unsigned)(q[2] >> 32), q3l = (unsigned)q[3], q3h = (unsigned)(q[3] >> 32);
    const unsigned q4l = (unsigned)q[4], q4h = (unsigned)(q[4] >> 32), q5l = (unsigned)q[5], q5h = (unsigned)(q[5] >> 32);
    asm volatile(
        // xor 1, transposing: even lanes keep the I of a pair, odd lanes its Q (a: what the neighbour needs, b: own)
        "v_cndmask_b32 %0, %23, %21, %33\n\t"  "v_cndmask_b32 %1, %24, %22, %33\n\t"
        "v_cndmask_b32 %2, %21, %23, %33\n\t"  "v_cndmask_b32 %3, %22, %24, %33\n\t"
        "v_cndmask_b32 %4, %27, %25, %33\n\t"  "v_cndmask_b32 %5, %28, %26, %33\n\t"
        "v_cndmask_b32 %6, %25, %27, %33\n\t"  "v_cndmask_b32 %7, %26, %28, %33\n\t"
        "v_cndmask_b32 %8, %31, %29, %33\n\t"  "v_cndmask_b32 %9, %32, %30, %33\n\t"
        "v_cndmask_b32 %10, %29, %31, %33\n\t" "v_cndmask_b32 %11, %30, %32, %33\n\t"
        "v_add_co_u32_dpp %12, vcc, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %13, vcc, %1, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %14, vcc, %4, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %15, vcc, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %16, vcc, %8, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %17, vcc, %9, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        // xor 2: (prompt, early) transposing by bit 1 of the lane; late plainly
        "v_cndmask_b32 %18, %14, %12, %34\n\t" "v_cndmask_b32 %19, %15, %13, %34\n\t"
        "v_cndmask_b32 %20, %12, %14, %34\n\t" "v_cndmask_b32 %0, %13, %15, %34\n\t"
        "v_add_co_u32_dpp %2, vcc, %16, %16 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %3, vcc, %17, %17, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %4, vcc, %18, %20 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %5, vcc, %19, %0, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        // rotations by 4 and 8 inside the row keep the low two lane bits
        "v_add_co_u32_dpp %6, vcc, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %7, vcc, %3, %3, vcc row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %8, vcc, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %9, vcc, %5, %5, vcc row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %16, vcc, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %17, vcc, %7, %7, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_co_u32_dpp %12, vcc, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_addc_co_u32_dpp %13, vcc, %9, %9, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(c0), "=&v"(c1),
          "=&v"(c2), "=&v"(c3), "=&v"(p0), "=&v"(p1), "=&v"(e0), "=&v"(e1), "=&v"(l0), "=&v"(l1), "=&v"(x0), "=&v"(x1),
          "=&v"(x2)
        : "v"(q0l), "v"(q0h), "v"(q1l), "v"(q1h), "v"(q2l), "v"(q2h), "v"(q3l), "v"(q3h), "v"(q4l), "v"(q4h), "v"(q5l),
          "v"(q5h), "s"(odd1), "s"(odd2)
        : "vcc");
    (void)x3;
    pe = ((unsigned long long)p1 << 32) | p0;
    l = ((unsigned long long)l1 << 32) | l0;
}

template <int SB>
__device__ __forceinline__ int t3_map_role(T3Shared& S, const int8_t* __restrict__ rec, long long rec_alloc, int ms,
                                           long long pos0, int unit, int set, int tid, unsigned long long* __restrict__ xbase,
                                           bool fast, double step_nom, double spacing, bool uns, bool prof_on,
                                           bool prof_any, bool wb_on, int* __restrict__ err) {
    static_assert(SB == 1, "one-byte samples");
    const int lane = tid & 63;
    const long long limit = rec_alloc - 16;                  // bytes: the last 16-byte word that may be loaded
    const int g = (tid & (T3_LANES - 1)) + unit * T3_LANES;  // the lane's group inside the block's aligned window
    const long long lane_off = (long long)g * 16;
    T2_FP_DECL
    T3_WB_DECL
    const bool tl_on = (blockIdx.x & 7) == 0 && blockIdx.x < 8 * T3_MAXP;   // (channel 0 of an 8-channel launch)
    (void)tl_on;
    (void)wb_on;
    (void)prof_on;
    (void)spacing;
    // The chips this lane can meet: kb = ceil(2 t) >> 1 >= floor(t) for the anchor's code phase t >= (16 g - 16) step - 0.05
    // (head in [0, 15], rem in [0, step), the code NCO within 0.4 % of its basis); eight sign bits from chip `ws` on.
    int ws;
    unsigned win;
    {
        const int kf = (int)floor((double)(16 * g - 16) * step_nom - 0.05);
        ws = kf < -1 ? -1 : kf;
        const int bi = ws + 1;                               // bit k + 1 of the packed table is chip k
        const unsigned lo = S.cbits[bi >> 5], hi = S.cbits[(bi >> 5) + 1];
        win = (unsigned)((((unsigned long long)hi << 32) | lo) >> (bi & 31)) & 0xFFu;
    }
    signed char* const scr = reinterpret_cast<signed char*>(S.scratch) + (tid & 255) * 16;   // the lane's 16 bytes in LDS
    signed char* const rawb = reinterpret_cast<signed char*>(S.raw) + (tid & 255) * 16;      // ... as loaded
    const unsigned long long odd1 = 0xAAAAAAAAAAAAAAAAull, odd2 = 0xCCCCCCCCCCCCCCCCull;       // lanes with bit 0 / bit 1 set
    double magic = T2_MAGIC;
    T2_PIN(magic);
    const double step_max = 0.5 / 18.0;
    // ---- what the speculative pass of a block leaves for its final pass ----
    double tc = 0.0, ts = 0.0, t1c = 0.0, t1s = 0.0;         // moments of ALL samples (cos, sin parts; order 0 and 1)
    double fc = 0.0, fs_ = 0.0, f1c = 0.0, f1s = 0.0;        // moments of the samples IN FRONT of the boundary
    double pmc = 0.0, pms = 0.0, pm1c = 0.0, pm1s = 0.0;     // added to the front when the boundary comes one sample EARLIER
    double ppc = 0.0, pps = 0.0, pp1c = 0.0, pp1s = 0.0;     // ... one sample LATER
    double alE = 0.0, alP = 0.0, alL = 0.0;                  // arm a's sum = al_a (all) + be_a (front)
    double beE = 0.0, beP = 0.0, beL = 0.0;
    double dfr = 0.0;                // 10.625 - (mean of (b - 7.5)^2 over the front's samples) / 2: see the final pass
    double khd = 0.0;                // the half-chip boundary the lane follows
    double mid = 0.0;                // floor(u) + 0.5 of the pass
    double im1d = 0.0;               // (double)(ilo - 1): the anchor sample of the boundary position
    int i0 = 0;                      // block index of the lane's first sample (for the block start the pass assumed)
    int cut = 0;                     // block length the pass was cut for
    int bsw_s = 0;                   // the pass's switch sample: the lane's sample b lay in front of the boundary iff b < bsw_s
    long long spos = pos0;           // the block start the pass assumed
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;   // the lane's bytes, zeroed outside the block: part A of a pass -> its part B
    T2Raw<SB> nraw;                  // the lane's 16 samples of this set's NEXT block (requested one pass ahead)
    uint4 held = make_uint4(0u, 0u, 0u, 0u);   // ... taken out of nraw at the end of part B: part A then starts without a wait for
                                     // the loads in flight (the vector memory counter also counts the granule store just made)
    long long pos = pos0;            // first sample of the block the loop is at
#ifdef T3_COUNT   // (diagnosis) how often this wave leaves the fast path: [0] start / length mispredicted, [1] patched, [2] accumulated again, [3] direct
    int n_ev[4] = {0, 0, 0, 0};
#endif
    long long npos_pred = 0;         // first sample the bytes in nraw were requested for

    // The block that follows a block (POS, REM, BLK) when the code rate stays what it is: its first sample is exact, its
    // code phase exact up to the roundings of the reference's linspace (1e-12 chips), its length the one this phase gives
    // at this rate.
    auto advance = [&](long long& POS, double& REM, int& BLK, double STEP, double INV) {
        POS += BLK;
        REM = __builtin_fma((double)BLK, STEP, REM) - 1023.0;
        int nb = (int)ceil((1023.0 - REM) * INV);
        BLK = nb < 1 ? 1 : nb;
    };

    // THE SPECULATIVE PASS over the lane's 16 samples `rw` of a block that is taken to start at record sample POS with
    // prompt code phase REM, slope STEP (1 / STEP ~ INV) and length CUT, with the sample phasors of table CARR.  In two
    // parts, so that the block barrier can lie between them (the pass of block k + 2 starts right behind the final pass
    // of block k and ends in the first half of block k + 1's period):
    //   part A  the block's edges, the boundary, the chips, the two candidates; NEXT >= 0: the bytes of this set's next
    //           block are requested INTO `nraw` for a block start NEXT (rw may be nraw itself: its registers are free by
    //           then, so nothing is copied at the loop's end - a copy there would wait for the request just made);
    //   part B  the sixteen samples' moments.
    auto spec_a = [&](const uint4 rw, long long POS, double REM, double STEP, double INV, int CUT, const T3Carr& CARR,
                      long long NEXT) {
        const int head = (int)(POS & 15);
        i0 = g * 16 - head;
        const int ilo = i0 < 0 ? 0 : i0;
        const int dlo = ilo - i0;
        // (wave-uniform values as scalars: the final pass's test of them is then a scalar compare, not an exec mask)
        cut = __builtin_amdgcn_readfirstlane(CUT);
        spos = ((long long)__builtin_amdgcn_readfirstlane((int)(POS >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)POS);
        im1d = (double)(ilo - 1);
        // the lane's bytes with everything outside the block zeroed (only the block's first and last groups are cut)
        r0 = rw.x; r1 = rw.y; r2 = rw.z; r3 = rw.w;
        *reinterpret_cast<uint4*>(rawb) = make_uint4(r0, r1, r2, r3);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(i0 < 0 || i0 + 16 > CUT) != 0, 0)) {
            const int lo = i0 < 0 ? -i0 : 0;                 // first byte inside the block
            int hi = CUT - i0;                               // one past the last byte inside it
            hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
            const unsigned keep = ((1u << hi) - 1u) & ~((1u << lo) - 1u);   // one bit per byte
            auto expand = [](unsigned n4) { return (((n4 & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu; };   // 4 bits -> 4 bytes
            r0 &= expand(keep);
            r1 &= expand(keep >> 4);
            r2 &= expand(keep >> 8);
            r3 &= expand(keep >> 12);
        }
        *reinterpret_cast<uint4*>(scr) = make_uint4(r0, r1, r2, r3);
        // the half-chip boundary behind the anchor
        const double hm = __builtin_fma(im1d, STEP + STEP, REM + REM);
        khd = ceil(hm);
        const double fu = floor((khd - hm) * (0.5 * INV));
        const int khi = (int)khd;
        const int kb = khi >> 1;
        int bsw = dlo + (int)fu;                              // the lane's sample b lies behind the boundary iff b >= bsw
        bsw = bsw > 17 ? 17 : bsw;
        bsw_s = bsw;
        const int sh = kb - ws;
        unsigned bits;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(i0 < CUT && (unsigned)sh > 5u) != 0, 0)) {
            const int kk = (kb < -1 ? -1 : (kb > 1022 ? 1022 : kb)) + 1;
            const unsigned lo = S.cbits[kk >> 5], hi = S.cbits[(kk >> 5) + 1];
            bits = (unsigned)((((unsigned long long)hi << 32) | lo) >> (kk & 31)) & 7u;
        } else {
            bits = (win >> (sh & 7)) & 7u;
        }
        // chips as the high dwords of +-1.0, their differences (0, +-2), the arms' coefficients by the boundary's parity
        const unsigned c0h = 0x3FF00000u | (bits << 31), c1h = 0x3FF00000u | ((bits >> 1) << 31), c2h = 0x3FF00000u | ((bits >> 2) << 31);
        const double c0 = __hiloint2double((int)c0h, 0), c1 = __hiloint2double((int)c1h, 0), c2 = __hiloint2double((int)c2h, 0);
        const unsigned d01h = (unsigned)__double2hiint(c0 - c1), d12h = (unsigned)__double2hiint(c1 - c2);
        const unsigned odd = (unsigned)(-(khi & 1));          // all ones: an early / late boundary
        alE = __hiloint2double((int)((odd & c1h) | (~odd & c0h)), 0);
        alP = c1;
        alL = __hiloint2double((int)((odd & c2h) | (~odd & c1h)), 0);
        beE = __hiloint2double((int)(odd & d01h), 0);
        beP = __hiloint2double((int)(~odd & d01h), 0);
        beL = __hiloint2double((int)(odd & d12h), 0);
        {   // second order of the rate step, front part (the final pass says why): n samples b = 0 .. n - 1 in front
            const int nf = bsw > 16 ? 16 : (bsw < 1 ? 1 : bsw);
            const double n1 = (double)(nf - 1);
            const double m2f = __builtin_fma(n1, __builtin_fma((double)(2 * nf - 1), 1.0 / 6.0, -7.5), 56.25);
            dfr = __builtin_fma(-0.5, m2f, 10.625);
        }
        // this lane's sums can change only if the boundary is in reach of the group and the group has samples inside the
        // block (its candidates are zero otherwise); EVERY lane watches how far its boundary moves - a boundary that was
        // out of reach and comes two samples nearer has crossed one (code rate steps of a kHz-wide DLL on a channel without
        // signal: the wave then takes the direct path)
        const unsigned lv = (unsigned)(((bsw - 17) & (i0 - CUT)) >> 31);          // all ones: live
        mid = fu + 0.5;
        // the two candidates: sample bsw - 1 (leaves the front when the boundary comes one sample earlier) and sample
        // bsw (joins it when the boundary comes one later); bytes outside the block are zero already
        const int bm = bsw - 1;
        const int bmc = bm < 0 ? 0 : (bm > 15 ? 15 : bm), bpc = bsw > 15 ? 15 : bsw;
        const unsigned vm = lv & ~(unsigned)(bm >> 31), vp = lv & (unsigned)((bsw - 16) >> 31);
        const double2 Bm = CARR.T[T2_B + bmc], Bp = CARR.T[T2_B + bpc];
        int xmi, xpi;
        if (uns) {
            xmi = (int)reinterpret_cast<const unsigned char*>(scr)[bmc];
            xpi = (int)reinterpret_cast<const unsigned char*>(scr)[bpc];
        } else {
            xmi = (int)scr[bmc];
            xpi = (int)scr[bpc];
        }
        const double xm = -(double)(int)((unsigned)xmi & vm);
        const double xp = (double)(int)((unsigned)xpi & vp);
        if (NEXT >= 0) {
            // (everything that reads rw is above; the asm statement keeps it there)
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
            npos_pred = NEXT;
            nraw = t2_load<SB>(rec, (NEXT & ~15ll) + lane_off, limit);
        }
        pmc = xm * Bm.x;
        pms = xm * Bm.y;
        ppc = xp * Bp.x;
        pps = xp * Bp.y;
        const double dbm = (double)bmc - 7.5, dbp = (double)bpc - 7.5;
        pm1c = pmc * dbm;
        pm1s = pms * dbm;
        pp1c = ppc * dbp;
        pp1s = pps * dbp;
        T2STAMP(prof_on && NEXT >= 0, 1);   // edges, boundary, chips, candidates (waits for the bytes), next request made
    };
    auto spec_b = [&](const T3Carr& CARR, bool shadow) {
        // samples -> high dwords of their fp64 values (first: the final pass of the other set reads its parameters from LDS
        // right now, the sixteen reads below would queue in front of them)
        unsigned xh[16];
        {
            const unsigned rr[4] = {r0, r1, r2, r3};
            if (uns) {   // (a wave-uniform branch instead of a select per sample)
#pragma unroll
                for (int b = 0; b < 16; ++b) xh[b] = (unsigned)__double2hiint((double)(int)__builtin_amdgcn_ubfe(rr[b >> 2], 8 * (b & 3), 8));
            } else {
#pragma unroll
                for (int b = 0; b < 16; ++b) xh[b] = (unsigned)__double2hiint((double)(int)__builtin_amdgcn_sbfe((int)rr[b >> 2], 8 * (b & 3), 8));
            }
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) asm volatile("" : "+v"(xh[b]));
        // the block's 16 sample phasors (one batch of reads, one wait); lanes 0 .. 15 of the set keep a copy for the
        // final pass's patches (CARR is another block's table by then)
        t2_v2d Bt[16];
        {
            const unsigned ba = (unsigned)(unsigned long long)&CARR.T[T2_B];
#define T3_BLD(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(Bt[i]) : "v"(ba))
            T3_BLD(0); T3_BLD(1); T3_BLD(2); T3_BLD(3); T3_BLD(4); T3_BLD(5); T3_BLD(6); T3_BLD(7);
            T3_BLD(8); T3_BLD(9); T3_BLD(10); T3_BLD(11); T3_BLD(12); T3_BLD(13); T3_BLD(14); T3_BLD(15);
#undef T3_BLD
        }
        if ((tid & (T3_LANES - 1)) < 16) S.btab[set][tid & 15] = CARR.T[T2_B + (tid & 15)];
        const int fmask = (1 << bsw_s) - 1;                   // bit b set: sample b lies in front of the boundary
        double a0c = 0.0, a0s = 0.0, a1c = 0.0, a1s = 0.0, g0c = 0.0, g0s = 0.0, g1c = 0.0, g1s = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bt[0]), "+v"(Bt[1]), "+v"(Bt[2]), "+v"(Bt[3]), "+v"(Bt[4]), "+v"(Bt[5]), "+v"(Bt[6]),
                     "+v"(Bt[7]), "+v"(Bt[8]), "+v"(Bt[9]), "+v"(Bt[10]), "+v"(Bt[11]), "+v"(Bt[12]), "+v"(Bt[13]), "+v"(Bt[14]),
                     "+v"(Bt[15]));
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const double xs = __hiloint2double((int)xh[b], 0);
            unsigned mb_;                                     // all ones iff b < bsw (asm: the compiler turns the
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mb_) : "v"(fmask), "n"(b));   // builtin into a compare + select through VCC)
            const double xf = __hiloint2double((int)(xh[b] & mb_), 0);
            const double x1 = xs * ((double)b - 7.5);
            const double xf1 = xf * ((double)b - 7.5);
            a0c = __builtin_fma(xs, Bt[b].x, a0c);
            a0s = __builtin_fma(xs, Bt[b].y, a0s);
            a1c = __builtin_fma(x1, Bt[b].x, a1c);
            a1s = __builtin_fma(x1, Bt[b].y, a1s);
            g0c = __builtin_fma(xf, Bt[b].x, g0c);
            g0s = __builtin_fma(xf, Bt[b].y, g0s);
            g1c = __builtin_fma(xf1, Bt[b].x, g1c);
            g1s = __builtin_fma(xf1, Bt[b].y, g1s);
        }
        tc = a0c; ts = a0s; t1c = a1c; t1s = a1s;
        fc = g0c; fs_ = g0s; f1c = g1c; f1s = g1s;
        T2STAMP(prof_on && shadow, 3);   // accumulation
    };

    {
        // set 0 owns the even blocks: block 0's pass runs here.  Set 1 owns the odd ones: block 1's pass runs here as well,
        // with block 0's rates.  Either requests the bytes of its next block (2 / 3) where block 0's rates put it.
        const T3Code& C0 = S.code[0];
        T2_FP_TOP
        long long p = pos0;
        double rm = C0.start[1];
        int bl = C0.blk;
        const double st0 = C0.step, iv0 = C0.inv_step;
        if (set == 1) advance(p, rm, bl, st0, iv0);
        long long p2 = p;
        double rm2 = rm;
        int bl2 = bl;
        advance(p2, rm2, bl2, st0, iv0);
        advance(p2, rm2, bl2, st0, iv0);
        nraw = t2_load<SB>(rec, (p & ~15ll) + lane_off, limit);
        spec_a(nraw.a, p, rm, st0, iv0, bl, S.carr[0], p2);
        spec_b(S.carr[0], false);
        held = nraw.a;
    }
    bool pend_b = false;             // part B of this set's next pass is still to run
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T3Code& C = S.code[par];
        const T3Carr& CR = S.carr[par];
        // one batch of LDS reads: the chain part, the block's code phase, the rotated tables, the rate step
        const int4 hd = *reinterpret_cast<const int4*>(&C.blk);          // blk, stop, inv_step
        const double step = C.step;
        const double rem = C.start[1];                                   // (exact; posted a block ago)
        const double2 w1 = CR.T[T2_W1 + (tid & 15)], w2 = CR.T[T2_W2 + ((tid >> 4) & 7)], w3 = CR.T[T2_W3];
        const double2 ep = *reinterpret_cast<const double2*>(&CR.eps);   // eps, respec
        T2_USE(hd.z); T2_USE(step); T2_USE(rem); T2_USE(w1.x); T2_USE(w2.x); T2_USE(w3.x); T2_USE(ep.x);   // one batch, one wait
        if (__builtin_amdgcn_readfirstlane(hd.y)) break;      // (wave-uniform values as scalars: the loop stays a scalar loop)
        T2_FP_TOP
        const int blk = __builtin_amdgcn_readfirstlane(hd.x);
        const double inv_step = __hiloint2double(hd.w, hd.z);
        if ((it & 1) == set) {
        // ======== this set's block: the final pass (on the chain) ========
        __builtin_amdgcn_s_setprio(2);
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 1);   // parameters in registers
        double eps = ep.x;
        double gc, gs;
        bool direct = false;
        double aI0, aQ0, aI1, aQ1, aI2, aQ2;                  // the lane's six sums: arm 0 early, 1 prompt, 2 late
        // (either path below defines them; the statement costs nothing and spares the plain path six initialisations)
        asm volatile("" : "=v"(aI0), "=v"(aQ0), "=v"(aI1), "=v"(aQ1), "=v"(aI2), "=v"(aQ2));
        // ---- where the boundary is now: u' = (kh - h'(ilo - 1)) / (2 step'); r = u' - (floor(u) + 1/2) ----
        double r = (khd - __builtin_fma(im1d, step + step, rem + rem)) * (0.5 * inv_step) - mid;
        // ONE test for everything that is not the plain case (each term of it uniform or rare): the boundary within 1e-7
        // samples of a sample or moved by two; a block that does not lie where the pass put it; a code rate at which 18
        // samples span half a chip; tables the PLL wave evaluated in full
        bool plain;
        {
            const double a = fabs(r) - 0.5;                   // > 0: the boundary crossed a sample
            plain = !((fabs(a) < 1e-7) | (a > 1.0) | (blk != cut) | ((int)pos != (int)spos) | (step > step_max) | (ep.y != 0.0));
        }
        // group phasor G' = W1'[tid & 15] * W2'[(tid >> 4) & 7] * W3' (rotated tables) - HERE, between the compares above and
        // the branch on them below: a branch right behind its compare waits ~20 cycles for the result (43.16 -> 42.75 ms)
        {
            const double lc = __builtin_fma(w1.x, w2.x, -(w1.y * w2.y));
            const double ls = __builtin_fma(w1.x, w2.y, w1.y * w2.x);
            gc = __builtin_fma(lc, w3.x, -(ls * w3.y));
            gs = __builtin_fma(lc, w3.y, ls * w3.x);
            asm volatile("" : "+v"(gc), "+v"(gs));
        }
#ifdef T3_CHECK
        bool chk_moved = false;
        int chk_dpos = 0;
        const double chk_gc = gc, chk_gs = gs;
#endif
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 2);   // test + group phasor
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!plain) != 0, 0)) {
            unsigned long long me = 0, mb = 0;
            // The pass ran two blocks back: the block starts `dpos` samples from where it was put and is `blk - cut` samples
            // longer than it was cut (either in a few per cent of the blocks, a sample or two).  The samples' ABSOLUTE
            // positions, code phases and carrier phases do not depend on where the block was taken to start, so the lanes
            // keep what they have with their block indices moved - except the lanes that hold the block's first and last
            // samples, where [lo_s, hi_s) of the lane's bytes were inside the block for the pass and [lo_f, hi_f) are now:
            // a sample or two that cannot change sides of the lane's boundary are PATCHED in or out - raw byte, the pass's
            // sample phasor, the side the pass put it on; anything else accumulates again.
            const int dpos = (int)(pos - spos);
            if (__builtin_expect(blk != cut || dpos != 0, 0)) {   // (wave-uniform)
                const bool same_win = ((pos ^ spos) & ~15ll) == 0;
#ifdef T3_CHECK
                chk_moved = true;
                chk_dpos = dpos;
#endif
                const int i0n = i0 - dpos;
                int lo_s = -i0, hi_s = cut - i0, lo_f = -i0n, hi_f = blk - i0n;
                lo_s = lo_s < 0 ? 0 : (lo_s > 16 ? 16 : lo_s);
                hi_s = hi_s < 0 ? 0 : (hi_s > 16 ? 16 : hi_s);
                lo_f = lo_f < 0 ? 0 : (lo_f > 16 ? 16 : lo_f);
                hi_f = hi_f < 0 ? 0 : (hi_f > 16 ? 16 : hi_f);
                if (hi_s < lo_s) hi_s = lo_s;                 // (a lane beyond the block's end: an empty range)
                if (hi_f < lo_f) hi_f = lo_f;
                // runs of changed bytes: [a1, b1) at the block's start, [a2, b2) at its end
                const int a1 = lo_s < lo_f ? lo_s : lo_f, b1 = lo_s < lo_f ? lo_f : lo_s;
                const int a2 = hi_s < hi_f ? hi_s : hi_f, b2 = hi_s < hi_f ? hi_f : hi_s;
                const int cnt = (b1 - a1) + (b2 - a2);
                // Which side of the lane's boundary a changed sample lies on: a lane that follows the boundary AT the block's
                // start (half chip 0: the code phase wraps where the block starts, so every sample inside the block lies
                // behind it) or AT its end (half chip 2046: every sample inside lies in front) knows by construction, and
                // its boundary moved WITH the block - nothing of it crosses.  Any other lane keeps the side the pass saw
                // unless the sample is one of the two next to the boundary, which the rate step may move across.
                const bool edge0 = khd < 0.5, edge_e = khd > 2045.5 && khd < 2046.5;
                const bool st = b1 > a1, en = b2 > a2;
                const int ca = st ? a1 : a2, cb = st ? b1 : b2;
                // (edge0: ... unless the code phase the block starts with is not positive - the reference's own roundings
                // leave it at -1e-13 while the code NCO rests at its basis - and its first sample lies IN FRONT)
                const bool hard = cnt > 2 || (st && en) ||
                                  (cnt > 0 && (khd > 2046.5 || (edge0 && (bsw_s != lo_s || !(rem > 0.0))) ||
                                               (edge_e && bsw_s < hi_s) ||
                                               (!edge0 && !edge_e && bsw_s >= ca && bsw_s <= cb)));
                me = same_win ? __builtin_amdgcn_ballot_w64(hard) : ~0ull;
#ifdef T3_COUNT
                n_ev[0] += 1;
                if (me == 0 && __builtin_amdgcn_ballot_w64(cnt > 0) != 0) n_ev[1] += 1;
#endif
                if (me == 0) {
                    if (ep.y == 0.0 && __builtin_amdgcn_ballot_w64(cnt > 0) != 0) {
                        const double sgn = st ? (lo_f < lo_s ? 1.0 : -1.0) : (hi_f > hi_s ? 1.0 : -1.0);
#pragma unroll 1
                        for (int k = 0; k < 2; ++k) {
                            const bool act = k < cnt;
                            const int bb = act ? ca + k : 0;
                            const int xi = uns ? (int)reinterpret_cast<const unsigned char*>(rawb)[bb] : (int)rawb[bb];
                            const double x = act ? sgn * (double)xi : 0.0;
                            const double2 Bb = S.btab[set][bb];   // the table the pass ran with
                            const double wgt = (double)bb - 7.5;
                            const double c0 = x * Bb.x, s0 = x * Bb.y;
                            tc += c0;
                            ts += s0;
                            t1c = __builtin_fma(c0, wgt, t1c);
                            t1s = __builtin_fma(s0, wgt, t1s);
                            if (edge0 ? false : (edge_e ? true : bb < bsw_s)) {
                                fc += c0;
                                fs_ += s0;
                                f1c = __builtin_fma(c0, wgt, f1c);
                                f1s = __builtin_fma(s0, wgt, f1s);
                            }
                        }
                        if (cnt > 0 && (edge0 || edge_e)) {    // (nothing of this lane crosses: no candidates; the guard stays)
                            pmc = pms = pm1c = pm1s = 0.0;
                            ppc = pps = pp1c = pp1s = 0.0;
                        }
                    }
                    cut = blk;
                }
                // the lanes' block indices where the block really starts
                i0 = i0n;
                im1d -= (double)dpos;
                spos = pos;
            }
            r = (khd - __builtin_fma(im1d, step + step, rem + rem)) * (0.5 * inv_step) - mid;   // (the lanes' indices may have moved)
            const double a = fabs(r) - 0.5;
            const bool bad = (fabs(a) < 1e-7) || (a > 1.0);   // within 1e-7 samples of a sample / moved by two
            // (a code rate at which 18 samples span half a chip or more - a code NCO driven percents off by a kHz-wide DLL
            // on a channel without signal -: a group can meet two boundaries, every wave takes the direct path)
            mb = __builtin_amdgcn_ballot_w64(bad) | (step > step_max ? ~0ull : 0ull);
            if (__builtin_expect((mb | me) != 0 || ep.y != 0.0, 0)) {
#ifdef T3_COUNT
                n_ev[mb != 0 ? 3 : 2] += 1;
#endif
                if (mb != 0) {
                    direct = true;
                    // DIRECT: per-sample chips from the exact linspace ramps (tracking.py:166-188), this block's own
                    // sample phasors; the group phasor less the 7.5 samples of rate step the PLL wave turned W3 by
                    const T2Raw<SB> again = t2_load<SB>(rec, (pos & ~15ll) + lane_off, limit);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    int budget = 1 << 20;
                    while (lds_peek(&C.xflag) != it + 1 && --budget) __builtin_amdgcn_s_sleep(1);
                    const double stE = C.stp[0], stP = C.stp[1], stL = C.stp[2];
                    const double sE = C.start[0], sP = C.start[1], sL = C.start[2];
                    const double e75 = 7.5 * eps;
                    const double hc = __builtin_fma(e75, gs, gc), hs = __builtin_fma(-e75, gc, gs);   // G' (1 - j 7.5 eps)
                    double dI0 = 0.0, dQ0 = 0.0, dI1 = 0.0, dQ1 = 0.0, dI2 = 0.0, dQ2 = 0.0;
                    const int head = (int)(pos & 15);
                    const int j0 = g * 16 - head;
        #pragma unroll 1
                    for (int b = 0; b < 16; ++b) {
                        const int i = j0 + b;
                        if ((unsigned)i < (unsigned)blk) {
                            const double x = (double)t2_sample<SB>(again, b, uns);
                            const double2 Bb = CR.T[T2_B + b];
                            const double pc = __builtin_fma(hc, Bb.x, -(hs * Bb.y));
                            const double ps = __builtin_fma(hs, Bb.x, hc * Bb.y);
                            const double xs = ps * x, xc = pc * x;
                            const int kE = (int)ceil(ramp_at(i, stE, sE)), kP = (int)ceil(ramp_at(i, stP, sP)),
                                      kL = (int)ceil(ramp_at(i, stL, sL));
                            const double cE = (chip_bits2(S.cbits, kE) & 1u) ? -1.0 : 1.0;
                            const double cP = (chip_bits2(S.cbits, kP) & 1u) ? -1.0 : 1.0;
                            const double cL = (chip_bits2(S.cbits, kL) & 1u) ? -1.0 : 1.0;
                            dI0 = __builtin_fma(cE, xs, dI0);
                            dQ0 = __builtin_fma(cE, xc, dQ0);
                            dI1 = __builtin_fma(cP, xs, dI1);
                            dQ1 = __builtin_fma(cP, xc, dQ1);
                            dI2 = __builtin_fma(cL, xs, dI2);
                            dQ2 = __builtin_fma(cL, xc, dQ2);
                        }
                    }
                    aI0 = dI0; aQ0 = dQ0; aI1 = dI1; aQ1 = dQ1; aI2 = dI2; aQ2 = dQ2;
        #ifdef T3_CHECK_AT
                    if (it == T3_CHECK_AT && (blockIdx.x & 7) == 7) {
                        // (diagnosis) what the fused evaluation would have given this lane
                        const double dm = __hiloint2double((__double2hiint(r + 0.5) >> 31) & 0x3FF00000, 0);
                        const double dp = __hiloint2double((__double2hiint(0.5 - r) >> 31) & 0x3FF00000, 0);
                        const double z0c = __builtin_fma(dp, ppc, __builtin_fma(dm, pmc, fc));
                        const double z0s = __builtin_fma(dp, pps, __builtin_fma(dm, pms, fs_));
                        const double z1c = __builtin_fma(dp, pp1c, __builtin_fma(dm, pm1c, f1c));
                        const double z1s = __builtin_fma(dp, pp1s, __builtin_fma(dm, pm1s, f1s));
                        const double Fc = __builtin_fma(-eps, z1s, z0c), Fs = __builtin_fma(eps, z1c, z0s);
                        const double Tc = __builtin_fma(-eps, t1s, tc), Ts = __builtin_fma(eps, t1c, ts);
                        const double tI = __builtin_fma(gs, Tc, gc * Ts), fI = __builtin_fma(gs, Fc, gc * Fs);
                        const double fI0 = __builtin_fma(beE, fI, alE * tI), fI1 = __builtin_fma(beP, fI, alP * tI), fI2 = __builtin_fma(beL, fI, alL * tI);
                        if (fabs(fI0 - dI0) + fabs(fI1 - dI1) + fabs(fI2 - dI2) > 1e-3 || fabs(fabs(r) - 0.5) < 1e-3)
                            printf("[t3 at] unit %d tid %d i0 %d blk %d khd %.1f mid %.3f r %.9f bsw %d dm %.0f dp %.0f | fused E %.4f P %.4f L %.4f | direct E %.4f P %.4f L %.4f | al %.0f %.0f %.0f be %.0f %.0f %.0f | pm %.3f pp %.3f rem %.6e step %.12e\n",
                                   unit, tid, i0, blk, khd, mid, r, bsw_s, dm, dp, fI0, fI1, fI2, dI0, dI1, dI2, alE, alP, alL, beE, beP, beL, pmc, ppc, rem, step);
                    }
        #endif
                } else {
                    const T2Raw<SB> again = t2_load<SB>(rec, (pos & ~15ll) + lane_off, limit);   // (an L2 hit: read before)
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) HERE: left to the compiler, a path that does not read every
                                                          // register of `again` costs every block a full wait at the loop's top
                    // Accumulate again with the true start, code phase, rate and length and THIS block's own sample
                    // phasors.  The rotated tables carry the 7.5 samples of the two-block rate step for lanes whose
                    // moments stem from the older table (theta = 7.5 eps); these lanes' do not: G' e^{-j theta}, no
                    // first-order term, and the factor 1 + 10.625 eps^2 that the other lanes' second-order term leaves
                    // in their sums (the recorded sums are multiplied by its inverse, off the chain).
                    spec_a(again.a, pos, rem, step, inv_step, blk, CR, -1ll);
                    spec_b(CR, false);
                    r = (khd - __builtin_fma(im1d, step + step, rem + rem)) * (0.5 * inv_step) - mid;   // (no move left)
                    if (ep.y == 0.0) {
                        const double th = 7.5 * eps;
                        const double k2 = __builtin_fma(-17.5 * eps, eps, 1.0);   // 1 - theta^2 / 2 + 10.625 eps^2
                        const double hc = __builtin_fma(th, gs, k2 * gc), hs = __builtin_fma(-th, gc, k2 * gs);
                        gc = hc;
                        gs = hs;
                        eps = 0.0;
                    }
                }
            }
        }
        T2STAMP(prof_on, 2);   // group phasor, where the block really lies, the boundary's guard
        if (!direct) {
            // 1.0 where the boundary crossed a sample downwards / upwards (arithmetic on the sign bit; the candidates of a
            // lane that cannot change are zero, whatever r is)
            const double dm = __hiloint2double((__double2hiint(r + 0.5) >> 31) & 0x3FF00000, 0);
            const double dp = __hiloint2double((__double2hiint(0.5 - r) >> 31) & 0x3FF00000, 0);
            // ---- the front's patch, first-order carrier correction of both sets ----
            const double z0c = __builtin_fma(dp, ppc, __builtin_fma(dm, pmc, fc));
            const double z0s = __builtin_fma(dp, pps, __builtin_fma(dm, pms, fs_));
            const double z1c = __builtin_fma(dp, pp1c, __builtin_fma(dm, pm1c, f1c));
            const double z1s = __builtin_fma(dp, pp1s, __builtin_fma(dm, pm1s, f1s));
            // SECOND ORDER: sum x_b B_b e^{j eps (b - 7.5)} lacks - eps^2 / 2 sum x_b B_b (b - 7.5)^2 here, which for a signal
            // that is constant over the samples summed is (eps^2 / 2) (mean of (b - 7.5)^2 over them) times the sum: 10.625
            // eps^2 for all sixteen - the factor the RECORDED sums are divided by, off the chain - but another figure for the n
            // samples in front of the boundary.  With rate steps over TWO blocks the difference (1e-12 of an arm's sum, slowly
            // varying with where the boundaries lie in the groups) moved the code NCO by 1e-12 Hz for thousands of blocks on
            // end; the front gets the factor that leaves it with the same 1 + 10.625 eps^2 as everything else.
            const double ffr = __builtin_fma(eps * eps, dfr, 1.0);
            const double Fc = __builtin_fma(-eps, z1s, z0c) * ffr, Fs = __builtin_fma(eps, z1c, z0s) * ffr;
            const double Tc = __builtin_fma(-eps, t1s, tc), Ts = __builtin_fma(eps, t1c, ts);
            // rotate by the group phasor: cos part -> Q, sin part -> I (tracking.py:205-207)
            const double tQ = __builtin_fma(gc, Tc, -(gs * Ts)), tI = __builtin_fma(gs, Tc, gc * Ts);
            const double fQ = __builtin_fma(gc, Fc, -(gs * Fs)), fI = __builtin_fma(gs, Fc, gc * Fs);
            aI0 = __builtin_fma(beE, fI, alE * tI);
            aQ0 = __builtin_fma(beE, fQ, alE * tQ);
            aI1 = __builtin_fma(beP, fI, alP * tI);
            aQ1 = __builtin_fma(beP, fQ, alP * tQ);
            aI2 = __builtin_fma(beL, fI, alL * tI);
            aQ2 = __builtin_fma(beL, fQ, alL * tQ);
        }
#ifdef T3_CHECK
        // (diagnosis build) every lane's six sums against the direct evaluation: the first blocks, and every block whose
        // start or length the pass had mispredicted
#ifndef T3_CHECK_ALL
#define T3_CHECK_ALL 0
#endif
        if ((!direct || T3_CHECK_ALL) && (it < 6 || chk_moved || T3_CHECK_ALL)) {
            const T2Raw<SB> again = t2_load<SB>(rec, (pos & ~15ll) + lane_off, limit);
            int budget = 1 << 20;
            while (lds_peek(&C.xflag) != it + 1 && --budget) __builtin_amdgcn_s_sleep(1);
            const double stE = C.stp[0], stP = C.stp[1], stL = C.stp[2];
            const double sE = C.start[0], sP = C.start[1], sL = C.start[2];
            const double e75 = 7.5 * ep.x;
            const double gc0 = chk_gc, gs0 = chk_gs;
            const double hc = __builtin_fma(e75, gs0, gc0), hs = __builtin_fma(-e75, gc0, gs0);
            double dI0 = 0.0, dQ0 = 0.0, dI1 = 0.0, dQ1 = 0.0, dI2 = 0.0, dQ2 = 0.0;
            const int j0 = g * 16 - (int)(pos & 15);
            for (int b = 0; b < 16; ++b) {
                const int i = j0 + b;
                if ((unsigned)i < (unsigned)blk) {
                    const double x = (double)t2_sample<SB>(again, b, uns);
                    const double2 Bb = CR.T[T2_B + b];
                    const double pc = __builtin_fma(hc, Bb.x, -(hs * Bb.y));
                    const double ps = __builtin_fma(hs, Bb.x, hc * Bb.y);
                    const double xs = ps * x, xc = pc * x;
                    const int kE = (int)ceil(ramp_at(i, stE, sE)), kP = (int)ceil(ramp_at(i, stP, sP)), kL = (int)ceil(ramp_at(i, stL, sL));
                    const double cE = (chip_bits2(S.cbits, kE) & 1u) ? -1.0 : 1.0;
                    const double cP = (chip_bits2(S.cbits, kP) & 1u) ? -1.0 : 1.0;
                    const double cL = (chip_bits2(S.cbits, kL) & 1u) ? -1.0 : 1.0;
                    dI0 = __builtin_fma(cE, xs, dI0); dQ0 = __builtin_fma(cE, xc, dQ0);
                    dI1 = __builtin_fma(cP, xs, dI1); dQ1 = __builtin_fma(cP, xc, dQ1);
                    dI2 = __builtin_fma(cL, xs, dI2); dQ2 = __builtin_fma(cL, xc, dQ2);
                }
            }
            const double e_ = fabs(aI0 - dI0) + fabs(aQ0 - dQ0) + fabs(aI1 - dI1) + fabs(aQ1 - dQ1) + fabs(aI2 - dI2) + fabs(aQ2 - dQ2);
#ifdef T3_CHECK_AT
            if (it == T3_CHECK_AT && (blockIdx.x & 7) == 7) {
                double sa = aI1, sd = dI1;
                for (int o = 32; o >= 1; o >>= 1) { sa += __shfl_xor(sa, o); sd += __shfl_xor(sd, o); }
                if (lane == 0) printf("[t3 at] unit %2d wave %d: prompt I of the wave %.4f, direct %.4f, direct flag %d blk %d cut %d\n", unit, tid >> 6, sa, sd, (int)direct, blk, cut);
            }
#endif
            if (e_ > 1e-3)
                printf("[t3 check] ch %d it %d unit %d tid %d i0 %d cut %d blk %d dpos %d khd %.1f mid %.3f r %.6f bsw %d | E %.4f/%.4f P %.4f/%.4f L %.4f/%.4f | al %.0f %.0f %.0f be %.0f %.0f %.0f | t %.3f f %.3f eps %.3e\n",
                       (int)(blockIdx.x & 7), it, unit, tid, i0, cut, blk, chk_dpos, khd, mid, r, bsw_s, aI0, dI0, aI1, dI1, aI2, dI2, alE, alP, alL, beE, beP, beL, tc, fc, ep.x);
        }
#endif
        T2STAMP(prof_on, 4);   // patch, first-order correction, rotation, three arms
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 3);
        // fixed point: the raw bits of fma(a, 2^28, 1.5 2^52) are bias + round(a 2^28); sums of them carry the sum of the
        // integers in their low 48 bits whatever the biases add up to
        const double lane_fix = uns ? T3_FIX * 0.5 : T3_FIX;
        constexpr unsigned long long res_mask = 0xFFFFFFFFFFFFull;
        unsigned long long q[6];
        {
            double t_[6];
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[0]) : "v"(aI1), "s"(lane_fix), "v"(magic));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[1]) : "v"(aQ1), "s"(lane_fix), "v"(magic));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[2]) : "v"(aI0), "s"(lane_fix), "v"(magic));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[3]) : "v"(aQ0), "s"(lane_fix), "v"(magic));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[4]) : "v"(aI2), "s"(lane_fix), "v"(magic));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t_[5]) : "v"(aQ2), "s"(lane_fix), "v"(magic));
#pragma unroll
            for (int k = 0; k < 6; ++k) q[k] = (unsigned long long)__double_as_longlong(t_[k]);
        }
        // transposing reduction inside each row of 16 lanes; exchange order I_P Q_P I_E Q_E I_L Q_L
        unsigned long long vpe, vl;
        t3_reduce6(q, odd1, odd2, vpe, vl);
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 4);   // reduced inside the rows
        {
            const int rl = lane & 15;
            if (rl < 6) {
                const int word = rl < 4 ? rl : 4 + (rl & 1);
                const unsigned long long mine = ((rl < 4 ? vpe : vl) & res_mask) | (1ull << 56);
                const unsigned long long prev = atomicAdd(&S.acc[par][word], mine);
                if ((prev >> 56) == 7ull) {
                    // the eighth arrival (2 waves x 4 rows): this lane holds the member's total of its word
                    const unsigned long long tot = prev + mine;
                    S.acc[par][word] = 0ull;
                    const unsigned long long gran = ((unsigned long long)((unsigned)(it + 1) & 0xFFFFu) << 48) | (tot & 0xFFFFFFFFFFFFull);
                    granule_store(xbase + ((((par * 3 + (word >> 1)) * T3_XLINE + unit) << 1) | (word & 1)), gran, fast);
#ifdef T3_TIMELINE
                    if (word == 0) S.tpub[par] = (long long)__builtin_amdgcn_s_memtime();
#else
                    if (prof_any && word == 0) S.tpub[par] = (long long)__builtin_amdgcn_s_memtime();
#endif
                }
            }
        }
        T2STAMP(prof_on, 5);   // published (or handed to the lanes that publish)
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 5);
        __builtin_amdgcn_s_setprio(T3_PRIO_SPEC);
        if (it + 2 < ms) {
            // ======== part A of the pass of this set's NEXT block, it + 2, with this block's rates (in the shadow of this
            // block's exchange and loop filter).  Its first sample is pos + blk + (the length block it + 1 will most likely
            // have), its code phase follows from this block's; the final pass sees the exact values (its guard is 2.7e-9).
            long long p2 = pos;
            double rm2 = rem;
            int bl2 = blk;
            advance(p2, rm2, bl2, step, inv_step);
            advance(p2, rm2, bl2, step, inv_step);
            if (__builtin_expect((p2 & ~15ll) != (npos_pred & ~15ll), 0)) {    // (the bytes were requested for a block start
                nraw = t2_load<SB>(rec, (p2 & ~15ll) + lane_off, limit);       //  in another 16-byte window)
                held = nraw.a;
            }
            long long p4 = p2;
            double rm4 = rm2;
            int bl4 = bl2;
            advance(p4, rm4, bl4, step, inv_step);
            advance(p4, rm4, bl4, step, inv_step);
            spec_a(held, p2, rm2, step, inv_step, bl2, CR, p4);
            pend_b = true;
        }
        } else if (pend_b) {
#if T3_NOP_SPEC > 0
            asm volatile(".rept " T3_STR(T3_NOP_SPEC) "\n\ts_nop 3\n\t.endr");   // (diagnosis) part B starts 16 cycles later each
#endif
            // ======== part B of that pass: the table is the one part A ran with - the block before this one's
            spec_b(S.carr[par ^ 1], true);
            pend_b = false;
            held = nraw.a;           // (requested a period ago: no wait to speak of)
            asm volatile("" : "+v"(held.x), "+v"(held.y), "+v"(held.z), "+v"(held.w));
        }
        pos += blk;
        T2STAMP(prof_on, 6);   // next block's speculative pass
        __builtin_amdgcn_sched_barrier(0);
        T3_TL(T3_TL_MAP, tl_on, tid >> 6, 6);   // at the barrier
        T3_WB(wb_on);
        __builtin_amdgcn_sched_barrier(0);
        T2STAMP(prof_on, 7);   // waiting for the loop filter
    }
    T2_FP_PRINT(prof_on && lane == 0, 0, 8)
    T3_WB_PRINT(wb_on, "map", ms)
#ifdef T3_COUNT
    if (lane == 0 && (blockIdx.x & 7) == 0 && blockIdx.x < 8 * 20 && (n_ev[0] | n_ev[2] | n_ev[3]))
        printf("[t3 count] unit %2d wave %d: %d blocks, start / length mispredicted %d, patched %d, accumulated again %d, direct %d\n", unit, tid >> 6, it, n_ev[0], n_ev[1], n_ev[2], n_ev[3]);
#endif
    return it;
}

// SEVERAL LOOKS IN FLIGHT.  A granule is seen one round trip after the load that finds it left, so with one load at a time the
// sums wait half a round trip on average for the next load to leave; two loads half a round trip apart halve that.  The
// load that is still in flight when the other one has found the sums lands LATER, in the middle of the loop filter: its
// destination must be a register the compiler never allocates.  So the whole poll is one asm statement on the physical
// registers v[244:255], which nothing else in this kernel uses (the kernel needs ~220; tests/test_cabi_and_host.py checks
// the disassembly), and the next poll starts by waiting for what the last one left.
// Returns the number of rounds left (0: nothing found in `rounds` rounds - the caller looks at the abort word and polls
// again); x: the granule of every active lane.  Call with the lanes that poll as the active lanes.
#ifndef T3_POLL_GAP
#define T3_POLL_GAP 5              // (the round-4 DLL poll's gap between its two pairs of loads; unused since round 5)
#endif
#ifndef T3_POLL_GAP3
#define T3_POLL_GAP3 2             // ... between the PLL wave's three loads
#endif
#ifndef T3_POLL2
#define T3_POLL2 3                 // bit 0: the PLL wave, bit 1: the DLL wave (0: one load at a time, for comparison)
#endif
// s_sleep units (64 cycles) the filter waves rest behind the barrier before their off-chain work
#ifndef T3_SLEEP_PLL
#define T3_SLEEP_PLL 0
#endif
#ifndef T3_SLEEP_DLL
#define T3_SLEEP_DLL 0
#endif
#ifndef T3_NOP_PLL
#define T3_NOP_PLL 0
#endif
#ifndef T3_NOP_DLL
#define T3_NOP_DLL 0
#endif
#ifndef T3_NOP_SPEC
#define T3_NOP_SPEC 0
#endif
#ifndef T3_NOP_REC
#define T3_NOP_REC 0
#endif
#ifdef T3_ALIGN_POLL   // (diagnosis) the polls' loops start on a 64-byte line of the instruction cache
#define T3_POLL_ALIGN ".p2align 6\n"
#else
#define T3_POLL_ALIGN
#endif
__device__ __forceinline__ int t3_poll1(unsigned long long& x, const unsigned long long* p, unsigned long long tag, int rounds) {
    unsigned long long t;
    int left;
    // (THREE loads in flight here: the PLL wave's chain is the longer of the two filter waves')
    asm volatile(
        "s_waitcnt vmcnt(0)\n\t"
        "global_load_dwordx2 v[250:251], %[p], off sc1\n\t"
        "s_sleep " T3_STR(T3_POLL_GAP3) "\n\t"
        "global_load_dwordx2 v[252:253], %[p], off sc1\n\t"
        "s_sleep " T3_STR(T3_POLL_GAP3) "\n\t"
        "global_load_dwordx2 v[254:255], %[p], off sc1\n\t"
        "s_mov_b32 %[n], %[r]\n"
        T3_POLL_ALIGN
        "1:\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_lshrrev_b64 %[t], 48, v[250:251]\n\t"
        "v_cmp_eq_u64_e32 vcc, %[tag], %[t]\n\t"
        "s_cmp_eq_u64 vcc, exec\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dwordx2 v[250:251], %[p], off sc1\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_lshrrev_b64 %[t], 48, v[252:253]\n\t"
        "v_cmp_eq_u64_e32 vcc, %[tag], %[t]\n\t"
        "s_cmp_eq_u64 vcc, exec\n\t"
        "s_cbranch_scc1 3f\n\t"
        "global_load_dwordx2 v[252:253], %[p], off sc1\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_lshrrev_b64 %[t], 48, v[254:255]\n\t"
        "v_cmp_eq_u64_e32 vcc, %[tag], %[t]\n\t"
        "s_cmp_eq_u64 vcc, exec\n\t"
        "s_cbranch_scc1 5f\n\t"
        "global_load_dwordx2 v[254:255], %[p], off sc1\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch 4f\n"
        "2:\n\t"
        "v_mov_b64 %[x], v[250:251]\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b64 %[x], v[252:253]\n\t"
        "s_branch 4f\n"
        "5:\n\t"
        "v_mov_b64 %[x], v[254:255]\n"
        "4:\n"
        : [x] "+v"(x), [t] "=&v"(t), [n] "=&s"(left)
        : [p] "v"(p), [tag] "s"(tag), [r] "s"(rounds)
        : "vcc", "scc", "memory", "v250", "v251", "v252", "v253", "v254", "v255");
    return left;
}
// The DLL wave's poll: an arm's I and Q granules of a unit lie side by side and ONE 16-byte load per look fetches both
// (round 5; before: two 8-byte loads per look, two looks in flight - this wave found its sums 430 cycles after the PLL wave
// found its own and was the last at the barrier of every block, profiles/r05_trk_phase_profile.txt).  Three in flight.
#ifndef T3_POLL_GAPD
#define T3_POLL_GAPD T3_POLL_GAP3    // s_sleep units between the DLL wave's three loads
#endif
#define T3_TAGSEL 0x07060302u      // v_perm_b32: {upper half of the first source, upper half of the second}
__device__ __forceinline__ int t3_poll2(unsigned long long& x1, unsigned long long& x2, const unsigned long long* p1,
                                        const unsigned long long* p2, unsigned long long tag, int rounds) {
    (void)p2;
    unsigned t;
    int left;
    const unsigned tag2 = (unsigned)tag | ((unsigned)tag << 16);
#define T3_P2_CHECK(r0, r1, r3, lbl)                                   \
        "s_waitcnt vmcnt(2)\n\t"                                        \
        "v_perm_b32 %[t], v" #r3 ", v" #r1 ", %[sel]\n\t"               \
        "v_cmp_eq_u32_e32 vcc, %[tag2], %[t]\n\t"                       \
        "s_cmp_eq_u64 vcc, exec\n\t"                                    \
        "s_cbranch_scc1 " lbl "\n\t"                                    \
        "global_load_dwordx4 v[" #r0 ":" #r3 "], %[p], off sc1\n\t"
    asm volatile(
        "s_waitcnt vmcnt(0)\n\t"
        "global_load_dwordx4 v[244:247], %[p], off sc1\n\t"
        "s_sleep " T3_STR(T3_POLL_GAPD) "\n\t"
        "global_load_dwordx4 v[248:251], %[p], off sc1\n\t"
        "s_sleep " T3_STR(T3_POLL_GAPD) "\n\t"
        "global_load_dwordx4 v[252:255], %[p], off sc1\n\t"
        "s_mov_b32 %[n], %[r]\n"
        T3_POLL_ALIGN
        "1:\n\t"
        T3_P2_CHECK(244, 245, 247, "2f")
        T3_P2_CHECK(248, 249, 251, "3f")
        T3_P2_CHECK(252, 253, 255, "5f")
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch 4f\n"
        "2:\n\t"
        "v_mov_b64 %[x1], v[244:245]\n\t"
        "v_mov_b64 %[x2], v[246:247]\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b64 %[x1], v[248:249]\n\t"
        "v_mov_b64 %[x2], v[250:251]\n\t"
        "s_branch 4f\n"
        "5:\n\t"
        "v_mov_b64 %[x1], v[252:253]\n\t"
        "v_mov_b64 %[x2], v[254:255]\n"
        "4:\n"
        : [x1] "+v"(x1), [x2] "+v"(x2), [t] "=&v"(t), [n] "=&s"(left)
        : [p] "v"(p1), [tag2] "s"(tag2), [r] "s"(rounds), [sel] "s"(T3_TAGSEL)
        : "vcc", "scc", "memory", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254",
          "v255");
#undef T3_P2_CHECK
    return left;
}

// ================================ PLL (wave 4) ================================
// Lane = 32 word + unit polls that unit's granule of I_P (word 0) / Q_P (word 1).  The next block's tables are computed
// IN FULL for the current rate while the wave waits for the sums; the rate step then turns every entry (W3 by 7.5
// samples more: the map's moments are taken about the group's centre).
__device__ __forceinline__ int t3_pll_role(T3Shared& S, const TrkConst& K, const TrkChan& cc, int unit, bool owner, int lane,
                                           int P, int ch, unsigned long long* __restrict__ xbase, int* __restrict__ err,
                                           bool prof_on, long long* __restrict__ prof, bool wb_on) {
    (void)wb_on;
    // tracking.py:123-130
    long long acc_map = 0, acc_xch = 0, acc_flt = 0, t_top = 0, t_arr = 0;   // SGX_TRK_PROFILE=1: per-member phase times
    double carrBasis = cc.acquiredFreq;
    double remCarr = 0.0, w_cur = (cc.acquiredFreq * 2.0) * M_PI, oldCarrNco = 0.0, oldCarrErr = 0.0;
    double w_prev = w_cur;           // the rate of the block BEFORE the current one (block 0: its own): the next block's moments
                                     // were accumulated with that block's sample phasors
    const double two_pi = 2 * M_PI;
    double k_a = K.k_carr_a, k_b = K.k_carr_b, inv_2pi = K.inv_2pi, c_hi = K.inv_2pifs_hi, c_lo = K.inv_2pifs_lo,
           inv_fs = K.inv_fs, fs = K.fs;
    double k_ab = k_a + k_b;
    T2_PIN(k_ab);
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(inv_2pi); T2_PIN(c_hi); T2_PIN(c_lo); T2_PIN(inv_fs); T2_PIN(carrBasis);
    T2_PIN(fs);
    // the largest rate step the rotation takes: the farthest table entry is sample n_units * UNIT of the block
    double dw_max = SGX_ROT_MAX / ((double)(K.n_units * T3_UNIT) * K.inv_fs);
    T2_PIN(dw_max);
    // ... and the largest TWO-block step the map's expansion takes (eps = step / fs): what it leaves out, (7.5 eps)^2 times a
    // few tenths that depend on where the boundaries lie in the groups, went into the code NCO's integrator during the
    // pull-in (steps of 50 Hz and more) and stayed there - the code phase then drifts from the reference's by 1e-15 chips
    // per block for the rest of the run.  Beyond 20 Hz (1e-12 left out; 3.4 sigma of a locked 45 dB-Hz channel's steps) the
    // tables are evaluated in full and the map accumulates again, exactly.
#ifndef T3_DW2_HZ
#define T3_DW2_HZ 20.0
#endif
    double dw2_lim = 2.0 * M_PI * T3_DW2_HZ;
    T2_PIN(dw2_lim);
    const int ms = K.ms;
    SgxAtanCoef ak = sgx_atan_coef();
    SgxRotCoef rk = sgx_rot_coef();
    T2_PIN(ak.c0); T2_PIN(ak.c1); T2_PIN(ak.c2); T2_PIN(ak.c3); T2_PIN(ak.c4); T2_PIN(ak.c5); T2_PIN(ak.c6); T2_PIN(ak.c7); T2_PIN(ak.c8);
    T2_PIN(rk.s0); T2_PIN(rk.s1); T2_PIN(rk.s2); T2_PIN(rk.s3); T2_PIN(rk.s4); T2_PIN(rk.s5);
    T2_PIN(rk.c0); T2_PIN(rk.c1); T2_PIN(rk.c2); T2_PIN(rk.c3); T2_PIN(rk.c4); T2_PIN(rk.c5);
    double unfix = 1.0 / (K.uns != 0 ? T3_FIX * 0.5 : T3_FIX);
    T2_PIN(unfix);
    const bool w3 = lane >= 48;
    double ctr_fs = w3 ? 7.5 * K.inv_fs : 0.0;   // the map's moments are taken about the group's centre
    T2_PIN(ctr_fs);
    const bool mine = (lane & 31) < P;
    unsigned long long* const xabort = xbase + T3_XABORT;
    // record values of the block just finished (member 0), posted after the barrier: carrFreq I_P Q_P pllDiscr pllDiscrFilt
    double r_cf = 0.0, r_ip = 0.0, r_qp = 0.0, r_err = 0.0, r_nco = 0.0;
    double s2_blk = 1.0;             // 1 - 10.625 eps^2 of the block being processed
#ifdef T3_POLLSTAT
    long long ps_t = 0, ps_n = 0;
#endif
    T2_FP_DECL
    T3_WB_DECL
    const bool tl_on = ch == 0;
    (void)tl_on;
    (void)prof_on;
    __builtin_amdgcn_s_setprio(3);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T3Code& C = S.code[par];
        T3_TL(1, tl_on, 4, 0);   // released (the reference of every other stamp)
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
        if (__builtin_amdgcn_readfirstlane(hd.y)) break;
        T2_FP_TOP
        if (prof) t_top = (long long)__builtin_amdgcn_s_memtime();
        // ---- before the sums arrive ----
        // carrier phase at the end of this block (T5), exact remainder by FMA
        const int blk = __builtin_amdgcn_readfirstlane(hd.x);
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
        const int mi = t3_carr_mult(lane, unit, head_next);
        // the rate step dw turns the entry by dw m / fs radians, W3 by 7.5 samples of the TWO-block step dw + d1 more (the next
        // block's moments stem from the table of the block before this one): dw (m / fs + 7.5 / fs) + d1 7.5 / fs
        const double d1 = w_cur - w_prev;
        double mf = __builtin_fma((double)mi, inv_fs, ctr_fs);
        double ang0 = d1 * ctr_fs;
        double nco_base = __builtin_fma(-k_a, oldCarrErr, oldCarrNco);
        T2_PIN(nco_base);
        // ONE test behind the sums: the two-block step within dw2_lim.  With the previous step within it as well, this block's
        // is within twice that, which the rotation takes (dw_max is 54 Hz at the default front end)
        double dw_lim = (fabs(d1) <= dw2_lim && 2.0 * dw2_lim <= dw_max) ? dw2_lim : -1.0;
        T2_PIN(ang0); T2_PIN(dw_lim);
        double cs_p, sn_p;
        t2_carr_entry(c_hi, c_lo, inv_2pi, w_cur, rc, mi, w3, cs_p, sn_p);
        T2_PIN(cs_p); T2_PIN(sn_p); T2_PIN(mf); T2_PIN(rc);   // (keeps all of this ahead of the wait)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long* gp = xbase + ((((par * 3 + 0) * T3_XLINE + (lane & 31)) << 1) | (lane >> 5));   // (I_P, Q_P of a unit lie side by side)
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
        T3_TL(T3_TL_PLL, tl_on, 4, 1);   // poll entered
#if defined(T3_POLL2) && (T3_POLL2 & 1)
        for (;;) {
            int left = 1;
#ifdef T3_POLLSTAT
            const long long tq0 = (long long)__builtin_amdgcn_s_memtime();
#endif
            if (mine) left = t3_poll1(x, gp, tag, 16);
#ifdef T3_POLLSTAT
            ps_t += (long long)__builtin_amdgcn_s_memtime() - tq0;
            ps_n += 3 * (16 - __builtin_amdgcn_readfirstlane(left)) + 1;
#endif
            if (__builtin_amdgcn_readfirstlane(left) != 0) break;
            if ((budget -= 16) <= 0 || lds_peek(&S.flag[1]) != 0) {
                gave_up = true;
                break;
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
        T3_TL(T3_TL_PLL, tl_on, 4, 2);   // sums found
        T3_TL_SET(T3_TL_PLL, tl_on, 4, 6, lds_peek64(&S.tpub[par]));   // (this member's publish)
        if (prof) {
            t_arr = (long long)__builtin_amdgcn_s_memtime();
            const long long tp = lds_peek64(&S.tpub[par]);
#ifdef T3_PROF_PAR   // (diagnosis) phase times of the even (0) / odd (1) blocks only: figures are per TWO blocks then
            if ((it & 1) == T3_PROF_PAR)
#endif
            {
            acc_map += tp - t_top;       // barrier release -> this member's publish
            acc_xch += t_arr - tp;       // this member's publish -> every member's sums visible
            }
        }
        // sum of the units' payloads (integers: exact, order-free); lanes that poll nothing hold 0
        const double v = t3_sum48_half(x);   // rows 1 and 3: the sums over lanes 0..31 / 32..63 (in units of the fixed point)
        const double I_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16),
                                            __builtin_amdgcn_readlane(__double2loint(v), 16));
        const double Q_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48),
                                            __builtin_amdgcn_readlane(__double2loint(v), 48));
        // T7 PLL (tracking.py:223-235); atan(Q/I) / 2 / pi as one multiplication by RN(1 / (2 pi)) (1.5 ulp)
        const double carrError = sgx_atan_ratio_k(Q_P, I_P, ak) * inv_2pi;
        // carrNco = oldCarrNco + k_a (carrError - oldCarrErr) + carrError k_b, regrouped so that ONE operation follows the
        // discriminator: (oldCarrNco - k_a oldCarrErr) + carrError (k_a + k_b) - the first bracket is ready before the sums are
        // (a rounding of 1e-16 relative in another place than the reference's; the NCO is continuous in its inputs)
        const double carrNco = __builtin_fma(carrError, k_ab, nco_base);
        const double carrFreq = carrBasis + carrNco;
        const double w_new = (carrFreq * 2.0) * M_PI;
        oldCarrNco = carrNco;
        oldCarrErr = carrError;
        T2PROBE(prof_on, 9);   // discriminator + NCO
        T3_TL(T3_TL_PLL, tl_on, 4, 3);
        // carrier tables of the next block: the prepared entry turned by the rate step (exact: w_new - w_cur is)
        double eps_next = 0.0;
        if (it + 1 < ms) {
            const double dw = w_new - w_cur;
            double cs, sn, eps_n, respec_n;
            const double dw2 = dw + d1;              // w_new - w_prev
            {   // (the small rotation straight through, the test for it behind: no wait for the compare in front of the arithmetic)
                double es, ec;
                sgx_rot_small(__builtin_fma(dw, mf, ang0), rk, es, ec);
                cs = __builtin_fma(cs_p, ec, -(sn_p * es));
                sn = __builtin_fma(cs_p, es, sn_p * ec);
                eps_n = dw2 * inv_fs;
                respec_n = 0.0;
                asm volatile("" : "+v"(cs), "+v"(sn));
            }
            if (__builtin_expect(!(fabs(dw2) <= dw_lim), 0)) {
                t2_carr_entry(c_hi, c_lo, inv_2pi, w_new, rc, mi, w3, cs, sn);
                eps_n = 0.0;
                respec_n = 1.0;
            }
            S.carr[par ^ 1].T[lane] = make_double2(cs, sn);
            if (lane == 0) *reinterpret_cast<double2*>(&S.carr[par ^ 1].eps) = make_double2(eps_n, respec_n);
            eps_next = eps_n;
        }
        w_prev = w_cur;
        w_cur = w_new;
        remCarr = rc;
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            atomicCAS(err, 0, 1 + ch);
            __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        T2STAMP(prof_on, 10);  // carrier tables
        T3_TL(T3_TL_PLL, tl_on, 4, 4);   // at the barrier
        T3_WB(wb_on);
#if T3_SLEEP_PLL > 0
        __builtin_amdgcn_s_sleep(T3_SLEEP_PLL);    // the final pass has the SIMD to itself for a moment (see T3_SLEEP_PLL)
#endif
#if T3_NOP_PLL > 0
        asm volatile(".rept " T3_STR(T3_NOP_PLL) "\n\ts_nop 3\n\t.endr");   // 16 cycles each
#endif
        __builtin_amdgcn_s_setprio(T3_PRIO_PRE);   // what follows until the next poll is off the chain: the final pass (2) issues first,
                                         // the speculative pass (0) after it
        __builtin_amdgcn_sched_barrier(0);
        // (a unit's prompt sum beyond half the room of the 48-bit payload, see T3_FIX: the guard of offset-binary records;
        // for every record: without these few instructions behind the barrier the kernel is 0.6 ms SLOWER - the PLL wave's
        // work in front of its poll then starts a moment earlier, beside the final pass)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine && (unsigned)((int)(short)(unsigned short)(x >> 32) + 0x4000) > 0x7FFFu) != 0, 0)) {
            if (lane == 0) atomicOr(err, TRK_ERR_SCALE);
        }
        r_cf = carrFreq;                 // (the block's record values: nobody waits for these)
        r_ip = I_P * (s2_blk * unfix);
        r_qp = Q_P * (s2_blk * unfix);
        s2_blk = __builtin_fma(-10.625 * eps_next, eps_next, 1.0);
        r_err = carrError;
        r_nco = carrNco;
#ifdef T3_PROF_PAR
        if ((it & 1) == T3_PROF_PAR)
#endif
        if (prof) acc_flt += (long long)__builtin_amdgcn_s_memtime() - t_arr;   // sums visible -> barrier released
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
#ifndef T3_PROF_DLL
    if (prof && lane == 0) {
        prof[ch * T2_PROF_STRIDE + unit] = acc_map;
        prof[ch * T2_PROF_STRIDE + 64 + unit] = acc_xch;
        prof[ch * T2_PROF_STRIDE + 128 + unit] = acc_flt;
    }
#endif
    T2_FP_PRINT(prof_on && lane == 0, 8, 12)
    T3_WB_PRINT(wb_on, "pll", ms)
#ifdef T3_POLLSTAT
    if (lane == 0 && ch == 0 && (unit == 0 || unit == 10)) printf("[t3 pollstat] unit %d: %.1f cycles in the poll per block, %.2f loads checked per block -> %.0f cycles per load\n", unit, (double)ps_t / it, (double)ps_n / it, (double)ps_t / (double)ps_n);
#endif
    return it;
}

// ================================ DLL (wave 5) ================================
// Lanes work in parallel on the three ramps: lane & 3 = 0 early, 1 prompt, 2 late (3 repeats prompt); uniform results
// come from lane 1 / lane 0.  Two polls: lane = 32 w + unit reads word 2 + w (I_E, Q_E) and word 4 + w (I_L, Q_L).
struct T3DllConst {
    double fs, inv_fs, code_len, spacing;
    double inv_nb_lane;     // RN(1 / (nb_base + (lane & 7))): reciprocals of the plausible block lengths, one per lane
    int nb_base;
    long long rec_len;
};

__device__ __forceinline__ int t3_dll_role(T3Shared& S, const TrkConst& K, const T3DllConst& D, long long pos0, int blk0,
                                           int stop0, bool owner, int lane, int P, int ch,
                                           unsigned long long* __restrict__ xbase, int* __restrict__ err, bool prof_on,
                                           long long file_off, bool wb_on, long long* __restrict__ prof = nullptr,
                                           int unit = -1) {
    (void)wb_on; (void)prof; (void)unit;
    // tracking.py:114-121; block 0's chain part and ramp starts were posted before the loop
    double oldCodeNco = 0.0, oldCodeErr = 0.0;
    double k_a = K.k_code_a, k_b = K.k_code_b, basis = K.code_basis;
    double k_ab = k_a + k_b;
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(basis); T2_PIN(k_ab);
    const int ms = K.ms;
    double rem = 0.0, cf = K.code_basis;
    long long pos = pos0;
    int blk = blk0, stop = stop0;
    const int l4 = lane & 3;
    const double off = (l4 == 0) ? -D.spacing : ((l4 == 2) ? D.spacing : 0.0);   // rem - spc == rem + (-spc) exactly
    const int lim3 = K.n_units * T3_UNIT - 15;             // the longest block the units of the launch hold
    double unfix = 1.0 / (K.uns != 0 ? T3_FIX * 0.5 : T3_FIX);
    T2_PIN(unfix);
    const bool mine = (lane & 31) < P;
    unsigned long long* const xabort = xbase + T3_XABORT;
    // record values of the block just finished (member 0), posted after the barrier
    double r_ve = 0.0, r_vl = 0.0, r_cf = 0.0, r_err = 0.0, r_nco = 0.0;
    T2_FP_DECL
    T3_WB_DECL
    const bool tl_on = ch == 0;
    (void)tl_on;
    (void)prof_on;
#ifdef T3_PROF_DLL   // (diagnosis, SGX_TRK_PROFILE=1) the three phase times of the profile are the DLL wave's: release -> poll
    long long dp_in = 0, dp_wait = 0, dp_post = 0, dp_t0 = 0, dp_t1 = 0, dp_t2 = 0;   // entered -> sums found -> at the barrier
#endif
    __builtin_amdgcn_s_setprio(3);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        T3Code& C = S.code[par];
        T3Code& N = S.code[par ^ 1];
#ifdef T3_PROF_DLL
        if (prof) dp_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
        if (owner && it > 0) {
            double* R = S.rec[par ^ 1];
            // rows 1 / 3 hold the early / late arm's I (r_ve) and Q (r_vl) -> series 4 (I_E), 6 (Q_E), 5 (I_L), 8 (Q_L)
            if (lane == 16) { R[4] = r_ve; R[6] = r_vl; }
            if (lane == 48) { R[5] = r_ve; R[8] = r_vl; }
            if (lane == 0) {
                R[0] = (double)(pos + file_off);   // position after block it - 1 = first sample of block it
                R[1] = r_cf;
                R[9] = r_err;
                R[10] = r_nco;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) lds_poke(&S.rflag[1], it);
        }
        if (stop) break;
        T2_FP_TOP
        // (the sums of this block carry the factor 1 - 10.625 eps^2 in what is recorded)
        const double e_ = S.carr[par].eps;
        const double s2_blk = __builtin_fma(-10.625 * e_, e_, 1.0);
        // ---- before the sums arrive: the exact arithmetic of this block (T1, T3) and the next block's early part ----
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
        // code phase and first sample of the next block (T4) and that block's ramp starts
        const double t_last = ramp_at(blk - 1, stp, start);
        const double rn_lane = (t_last + step) - 1023.0;                            // meaningful in the prompt lane
        const double rem_next = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rn_lane), 1),
                                                 __builtin_amdgcn_readlane(__double2loint(rn_lane), 1));
        const long long pos_next = pos + blk;
        if (lane < 3) N.start[lane] = rem_next + off;
        if (lane == 0) N.pos = pos_next;
        double a_next = D.code_len - rem_next;                                      // (1023 - rem) of T1
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) lds_poke(&C.xflag, it + 1);
        // the longest next block the record (stop 1 beyond it) and the units of the launch (stop 3) hold: ONE compare on
        // the chain, which of the two it was is sorted out in the (rare) branch
        const long long room = D.rec_len - pos_next;
        const int lim1 = room > (long long)0x3FFFFFFF ? 0x3FFFFFFF : (room < 0 ? 0 : (int)room);
        unsigned lim = (unsigned)(lim1 < lim3 ? lim1 : lim3);
        double nco_base = __builtin_fma(-k_a, oldCodeErr, oldCodeNco);
        T2_PIN(a_next); T2_PIN(lim); T2_PIN(nco_base);
        __builtin_amdgcn_s_setprio(3);
        // lanes 0..31 follow the early arm, lanes 32..63 the late one: gp1 the I sums (words 2 | 4), gp2 the Q sums (3 | 5)
        const unsigned long long* gp1 = xbase + (((par * 3 + 1 + (lane >> 5)) * T3_XLINE + (lane & 31)) << 1);   // 16 bytes: I and Q
        const unsigned long long* gp2 = gp1 + 1;
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x1 = 0, x2 = 0, xa = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
#ifdef T3_PROF_DLL
        if (prof) dp_t1 = (long long)__builtin_amdgcn_s_memtime();
#endif
        T3_TL(T3_TL_DLL, tl_on, 5, 1);   // poll entered
#if defined(T3_POLL2) && (T3_POLL2 & 2)
        for (;;) {
            int left = 1;
            if (mine) left = t3_poll2(x1, x2, gp1, gp2, tag, 8);
            if (__builtin_amdgcn_readfirstlane(left) != 0) break;
            xa = __hip_atomic_load(xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (xa != 0 || (budget -= 8) <= 0) {
                gave_up = true;
                break;
            }
        }
#else
        for (;;) {
            if (mine) {
                x1 = __hip_atomic_load(gp1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x2 = __hip_atomic_load(gp2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__all(!mine || ((x1 >> 48) == tag && (x2 >> 48) == tag))) break;
            if ((--budget & 15) == 0) {
                xa = __hip_atomic_load(xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (xa != 0 || budget == 0) {
                    gave_up = true;
                    break;
                }
            }
        }
#endif
#ifdef T3_PROF_DLL
        if (prof) dp_t2 = (long long)__builtin_amdgcn_s_memtime();
#endif
        T2STAMP(prof_on, 12);  // waiting for the sums
        T3_TL(T3_TL_DLL, tl_on, 5, 2);   // sums found
        // T8 DLL (tracking.py:238-251).  Integer sums over the units (exact, order-free; lanes that poll nothing hold 0):
        // row 1 then holds the early arm's I (q1) and Q (q2), row 3 the late arm's; the two envelopes are ONE register
        double vi, vq;                       // row 1: I_E, Q_E; row 3: I_L, Q_L (in units of the fixed point)
        t3_sum48_half2(x1, x2, vi, vq);
        const double m2 = __builtin_fma(vq, vq, vi * vi);   // row 1: I_E^2 + Q_E^2, row 3: I_L^2 + Q_L^2
        const double mm = sgx_sqrt1_pos(m2);                // (two zero envelopes: NaN, as in the reference)
        const double mE = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mm), 16),
                                           __builtin_amdgcn_readlane(__double2loint(mm), 16));
        const double mL = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mm), 48),
                                           __builtin_amdgcn_readlane(__double2loint(mm), 48));
        const double codeError = sgx_div1(mE - mL, mE + mL);   // (E - L) / (E + L), uniform
        const double codeNco = __builtin_fma(codeError, k_ab, nco_base);   // (regrouped like the PLL wave's)
        const double cf_new = basis - codeNco;
        oldCodeNco = codeNco;
        oldCodeErr = codeError;
        T2PROBE(prof_on, 13);  // discriminator + NCO
        T3_TL(T3_TL_DLL, tl_on, 5, 3);
        // chain part of the next block: its length, the ramps' slope and the slope's reciprocal
        // (sgx_block_length's arithmetic with its guard - the quotient within 6 ulp of an integer, 1e-10 of the blocks - folded
        // into the ONE rare branch below: a compare in front of a branch costs the wave ~20 cycles of waiting)
        double step_a = cf_new * D.inv_fs, inv_step;
        int blk_n;
        bool near;
        {
            double y = SGX_RCP_SEED(step_a);
            y = __builtin_fma(y, __builtin_fma(-step_a, y, 1.0), y);
            const double q0 = a_next * y;
            const double q = __builtin_fma(__builtin_fma(-q0, step_a, a_next), y, q0);
            inv_step = y;
            const double c = ceil(q);
            const double tol = q * 1.4e-15;
            near = (c - q < tol) | (q - c + 1.0 < tol);
            blk_n = (int)c;
        }
        int stop_n = 0;
        if (__builtin_expect(near || (unsigned)(blk_n - 1) >= lim || gave_up, 0)) {      // (blk <= 0 wraps to a huge number)
            if (near) blk_n = (int)ceil(a_next / sgx_div_rn(cf_new, D.fs, D.inv_fs));
            if ((unsigned)(blk_n - 1) >= lim || gave_up) {
                stop_n = gave_up ? 2 : ((blk_n <= 0 || blk_n > lim1) ? 1 : 3);
                if (lane == 0 && stop_n == 3) {
                    atomicOr(err, TRK_ERR_RANGE);
                    atomicCAS(err + 1, 0, 1 + ch);
                }
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
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            if (xa == 0) {
                atomicCAS(err, 0, 1 + ch);
                __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        T2STAMP(prof_on, 14);  // next block's code parameters
#ifdef T3_PROF_DLL
        if (prof) {
#ifdef T3_PROF_PAR   // (even (0) / odd (1) blocks only: figures are per TWO blocks then)
            if ((it & 1) == T3_PROF_PAR)
#endif
            {
            dp_in += dp_t1 - dp_t0;
            dp_wait += dp_t2 - dp_t1;
            dp_post += (long long)__builtin_amdgcn_s_memtime() - dp_t2;
            }
        }
#endif
        T3_TL(T3_TL_DLL, tl_on, 5, 4);   // at the barrier
        T3_WB(wb_on);
#if T3_SLEEP_DLL > 0
        __builtin_amdgcn_s_sleep(T3_SLEEP_DLL);
#endif
#if T3_NOP_DLL > 0
        asm volatile(".rept " T3_STR(T3_NOP_DLL) "\n\ts_nop 3\n\t.endr");
#endif
        __builtin_amdgcn_s_setprio(T3_PRIO_PRE);
        __builtin_amdgcn_sched_barrier(0);
        r_ve = vi * (s2_blk * unfix);    // (the block's record values: nobody waits for these)
        r_vl = vq * (s2_blk * unfix);
        r_cf = cf_new;
        r_err = codeError;
        r_nco = codeNco;
        T2STAMP(prof_on, 15);
    }
    if (owner && it > 0 && it == ms) {
        double* R = S.rec[(it - 1) & 1];
        if (lane == 16) { R[4] = r_ve; R[6] = r_vl; }
        if (lane == 48) { R[5] = r_ve; R[8] = r_vl; }
        if (lane == 0) {
            R[0] = (double)(pos + file_off);
            R[1] = r_cf;
            R[9] = r_err;
            R[10] = r_nco;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) lds_poke(&S.rflag[1], it);
    }
    T2_FP_PRINT(prof_on && lane == 0, 12, 16)
#ifdef T3_PROF_DLL
    if (prof && lane == 0) {
        prof[ch * T2_PROF_STRIDE + unit] = dp_in;
        prof[ch * T2_PROF_STRIDE + 64 + unit] = dp_wait;
        prof[ch * T2_PROF_STRIDE + 128 + unit] = dp_post;
    }
#endif
    T3_WB_PRINT(wb_on, "dll", ms)
    return it;
}

// ================================ RECORD (wave 6) ================================
// Stores block k's 13 series values once both filter waves have posted them (rflag >= k + 1): one block behind.  A
// stalled filter wave is an error, never stale rows (the host repeats the launch).
__device__ __forceinline__ void t3_rec_store(T3Shared& S, int k, long long m, int lane, double* __restrict__ o,
                                             int* __restrict__ err, int ch) {
    int budget = 1 << 18;
    while ((lds_peek(&S.rflag[0]) < k + 1 || lds_peek(&S.rflag[1]) < k + 1) && --budget) __builtin_amdgcn_s_sleep(2);
    if (budget == 0) {
        if (lane == 0) atomicCAS(err, 0, 1 + ch);
        return;
    }
    if (lane < SGX_NUM_SERIES) o[lane * m + k] = S.rec[k & 1][lane];
}

// (a record that is still streaming in: see t2_rec_role in sgx_trk2.hip)
// PREFETCH: a unit reads 2 KB of every block - a new page for the CU's address translation and a miss all the way to
// HBM every time, 1.5-2 us from request to data, as long as a whole code period.  The map waves request a block's bytes
// one block ahead and cannot afford more registers in flight; this wave, idle otherwise, touches the unit's cache lines
// THREE blocks ahead (one dword per 128-byte line, one load instruction per block, result never read), so that the map
// waves' requests find the lines in L2 and the translation cached.
__device__ __forceinline__ int t3_rec_role(T3Shared& S, const TrkConst& K, const int8_t* __restrict__ rec, int unit, int ch,
                                           bool owner, int lane, double* __restrict__ o, int* __restrict__ err,
                                           unsigned long long mark_seen, bool wb_on) {
    (void)wb_on;
    const long long m = K.ms;
    const int ms = K.ms;
    const long long span = 2ll * K.n_units * T3_UNIT + 64;   // bytes: a block and the window of the prefetch behind it
    const long long limit = K.rec_alloc - 16;
    unsigned dummy = 0;
#ifdef T3_REC_LAT
    long long rl_acc = 0;
#endif
    T3_WB_DECL
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T3Code& C = S.code[par];
        if (C.stop) break;
#if T3_NOP_REC > 0
        asm volatile(".rept " T3_STR(T3_NOP_REC) "\n\ts_nop 3\n\t.endr");   // (diagnosis) the far prefetch leaves 16 cycles later each
#endif
        if (lane < 20) {
            long long a = ((C.pos + 6ll * C.blk) & ~127ll) + (long long)unit * T3_UNIT + 128ll * (lane - 1);
            a = a < 0 ? 0 : (a > limit ? limit : a);
            if (K.mark == nullptr || (unsigned long long)(a + 128) <= mark_seen)   // (a streaming record: only what is resident)
                asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(rec + a) : "memory");
        }
#ifndef T3_NO_GUARD
        if (K.mark != nullptr && K.uns == 0) {
            // THE SCALE GUARD (see T3_FIX) of a STREAMING int8 record (a resident one has been scanned once by the host,
            // sgx_trk.hip: if_mag_bound): the magnitudes of the unit's 2 048 bytes of THIS block (its aligned window: an
            // L2 hit, the map waves read them two blocks ago) - signed bytes as |(b ^ 0x80) - 0x80| by v_sad_u8, a DPP
            // reduction into lane 63.  On this wave because it has the time: the speculative pass has none to spare (15
            // instructions more in its part B: 42.8 -> 44.9 ms; here, for every record: + 0.17 ms).
            long long a = (C.pos & ~15ll) + (long long)unit * T3_UNIT + 32ll * lane;
            const long long top = limit - 16;
            a = a < 0 ? 0 : (a > top ? top : a);
            if (K.mark == nullptr || (unsigned long long)(a + 32) <= mark_seen) {
                const uint4 w0 = *reinterpret_cast<const uint4*>(rec + a);
                const uint4 w1 = *reinterpret_cast<const uint4*>(rec + a + 16);
                const unsigned bias = 0x80808080u;
                int mag = 0;
                mag = (int)__builtin_amdgcn_sad_u8(w0.x ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w0.y ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w0.z ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w0.w ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w1.x ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w1.y ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w1.z ^ bias, bias, (unsigned)mag);
                mag = (int)__builtin_amdgcn_sad_u8(w1.w ^ bias, bias, (unsigned)mag);
                mag += __builtin_amdgcn_update_dpp(0, mag, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                mag += __builtin_amdgcn_update_dpp(0, mag, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
                mag += __builtin_amdgcn_update_dpp(0, mag, 0x141, 0xF, 0xF, true);   // row_half_mirror
                mag += __builtin_amdgcn_update_dpp(0, mag, 0x140, 0xF, 0xF, true);   // row_mirror: every lane holds its row's sum
                mag += __builtin_amdgcn_update_dpp(0, mag, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
                mag += __builtin_amdgcn_update_dpp(0, mag, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
                if (__builtin_expect(__builtin_amdgcn_readlane(mag, 63) >= 131072, 0)) {
                    if (lane == 0) atomicOr(err, TRK_ERR_SCALE);
                }
            }
        }
#endif
#ifdef T3_REC_LAT   // (diagnosis) how long the far prefetch is under way: the CU returns vector loads in order
        {
            const long long t0_ = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy) : : "memory");
            rl_acc += (long long)__builtin_amdgcn_s_memtime() - t0_;
        }
#endif
        if (owner && it > 0) t3_rec_store(S, it - 1, m, lane, o, err, ch);
        {   // (a resident record: mark_seen is all ones and this returns at once)
            const long long need = C.pos + 4 * span;
            wait_mark(K.mark, need < K.rec_len ? need : K.rec_len, mark_seen, err, ch);
        }
        T3_WB(wb_on);
    }
#ifdef T3_REC_LAT
    if (lane == 0 && ch == 0 && (unit == 0 || unit == 10)) printf("[t3 rec] unit %d: the far prefetch returns %.0f cycles after its issue (mean of %d blocks)\n", unit, (double)rl_acc / it, it);
#endif
    T3_WB_PRINT(wb_on, "rec", ms)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy) : : "memory");   // (no request outlives the wave's registers)
    return it;
}

// A channel has P = K.split = K.n_units members, member = unit.
__global__ __launch_bounds__(T3_THREADS) void trk3_kernel(const int8_t* __restrict__ rec, const int8_t* __restrict__ codes,
                                                          const TrkChan* __restrict__ chans, double* __restrict__ out,
                                                          int* __restrict__ ms_done, TrkConst K,
                                                          long long* __restrict__ prof,
                                                          unsigned long long* __restrict__ xch, int* __restrict__ err) {
    __shared__ T3Shared S;
#ifdef T3_PAD   // (diagnosis) moves every instruction behind this point by 4 T3_PAD bytes: does the code's placement matter?
    asm volatile(".rept " T3_STR(T3_PAD) "\n\ts_nop 0\n\t.endr");
#endif
    const int P = K.split;
    const int bq = blockIdx.x >> 3, br = blockIdx.x & 7;
    const int ch = br + 8 * (bq / P);
    const int unit = bq % P;
    const bool owner = unit == 0;              // the member that records the channel's series
    if (ch >= K.n_ch) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const TrkChan cc = chans[ch];
    if (cc.prn == 0) {
        if (tid == 0 && owner) ms_done[ch] = 0;
        return;
    }
    unsigned long long* __restrict__ xbase = xch + (long long)ch * T3_XCH_STRIDE;   // granules, abort word, placement granules
    unsigned long long* const xabort = xbase + T3_XABORT;
    const bool prof_on = (owner && ch == 0 && prof != nullptr && (wave == 0 || wave == 4 || wave == 5));

    if (tid < 4) {
        S.flag[tid] = 0;
        S.rflag[tid] = 0;
    }

    if (tid < 16) S.acc[tid >> 3][tid & 7] = 0ull;
#ifdef T3_TIMELINE
    if (tid < 96) S.tl[tid >> 4][(tid >> 3) & 1][tid & 7] = 0ull;
#endif
    if (tid < 2) {
        S.code[tid].xflag = 0;
        S.carr[tid].eps = 0.0;
        S.carr[tid].respec = 0.0;
    }
    // sign bits of the extended code [c1022, c0 .. c1022, c0] (tracking.py:111): bit k + 1 is set where chip k is -1
    for (int base = wave * 64; base < 40 * 32; base += (T3_THREADS / 64) * 64) {   // (T3_THREADS / 64 = 7 waves)
        const int i = base + lane;
        const int j = (i - 2 + 2 * 1023) % 1023;             // chip k = i - 1 of the extended code is code[(k - 1) mod 1023]
        const bool neg = i < 1032 && codes[(cc.prn - 1) * 1023 + j] < 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(neg);
        if (lane == 0) {
            S.cbits[base >> 5] = (unsigned)m;
            S.cbits[(base >> 5) + 1] = (unsigned)(m >> 32);
        }
    }
    __syncthreads();
    // ---- placement: are all members of the channel on one XCD (one L2)?  Then the exchange may stay in that L2.
    if (wave == 4) {
        unsigned long long* pl = xbase + T3_XPLACE;
        const unsigned me = xcc_id();
        if (lane == 0) __hip_atomic_store(pl + unit, 0xC0DE000000000000ull | me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long x = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
        for (;;) {
            if (lane < P) x = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane >= P || (x >> 48) == 0xC0DE;
            if (__all(ok)) break;
            if (--budget == 0) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const bool same = __all(lane >= P || (unsigned)(x & 0xF) == me);
        if (lane == 0) {
            S.flag[0] = (same && !gave_up && K.fast_xcd != 0) ? 1 : 0;
            if (gave_up) {   // a member is not resident: give the channel up at once (the host repeats with split 1)
                S.flag[1] = 1;
                atomicCAS(err, 0, 1 + ch);
                __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    const bool fast = S.flag[0] != 0;
    const bool dead = S.flag[1] != 0;

    // block 0 parameters (tracking.py:114-130): chain part and ramp starts
    T3DllConst D;
    int blk0 = 0, stop0 = 0;
    if (wave == 5) {
        D.fs = K.fs;
        D.inv_fs = K.inv_fs;
        D.code_len = K.code_len;
        D.spacing = K.spacing;
        D.inv_nb_lane = 1.0 / (double)(K.nb_base + (lane & 7));
        D.nb_base = K.nb_base;
        D.rec_len = K.rec_len;
        double step_a, inv_step;
        blk0 = sgx_block_length(K.code_len - 0.0, K.code_basis, D.fs, D.inv_fs, step_a, inv_step);
        const int lim3 = K.n_units * T3_UNIT - 15;
        stop0 = dead ? 2 : ((blk0 <= 0 || cc.pos0 + blk0 > D.rec_len) ? 1 : ((blk0 > lim3) ? 3 : 0));
        const double off = ((lane & 3) == 0) ? -K.spacing : (((lane & 3) == 2) ? K.spacing : 0.0);
        if (lane < 3) S.code[0].start[lane] = 0.0 + off;
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
                      t3_carr_mult(lane, unit, (int)(cc.pos0 & 15)), lane >= 48, cs, sn);
        S.carr[0].T[lane] = make_double2(cs, sn);
    }
    // a streaming record: block 0 and the requests made for blocks 1 and 2 must be resident before the first loads
    unsigned long long mark_seen = K.mark ? 0ull : ~0ull;
    if (wave == 6 && K.mark) {
        const long long need = cc.pos0 + 4 * (2ll * K.n_units * T3_UNIT + 64);
        wait_mark(K.mark, need < K.rec_len ? need : K.rec_len, mark_seen, err, ch);
    }
    __syncthreads();

    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    const bool wb_on = T3_WB_ON(unit, ch);
    int done;
    if (wave < 4)
        done = t3_map_role<1>(S, rec, K.rec_alloc, K.ms, cc.pos0, unit, wave >> 1, tid, xbase, fast, K.code_basis / K.fs,
                              K.spacing, K.uns != 0, prof_on, prof != nullptr, wb_on, err);
    else if (wave == 4)
        done = t3_pll_role(S, K, cc, unit, owner, lane, P, ch, xbase, err, prof_on, prof, wb_on);
    else if (wave == 5)
        done = t3_dll_role(S, K, D, cc.pos0, blk0, stop0, owner, lane, P, ch, xbase, err, prof_on, K.file_off, wb_on, prof, unit);
    else
        done = t3_rec_role(S, K, rec, unit, ch, owner, lane, o, err, mark_seen, wb_on);

#ifdef T3_TIMELINE
    __syncthreads();
    if (ch == 0 && tid < 96 && done > T3_TL_FROM + 2 && S.tl[tid >> 4][(tid >> 3) & 1][tid & 7] != 0ull)
        printf("[t3 tl] unit %d wave %d par %d ev %d mean %.1f\n", unit, tid >> 4, (tid >> 3) & 1, tid & 7,
               (double)S.tl[tid >> 4][(tid >> 3) & 1][tid & 7] / (0.5 * (double)(done - T3_TL_FROM)));
#endif
    // a channel that was given up reports the blocks completed before the abort
    const bool aborted = S.code[done & 1].stop == 2;
    if (wave == 6 && owner && done > 0 && !aborted) t3_rec_store(S, done - 1, (long long)K.ms, lane, o, err, ch);
    if (aborted && done > 0) done -= 1;
    if (tid == 0 && owner) ms_done[ch] = done;
}

// n_blocks = 8-padded channels x K.split workgroups; lds_pad: extra dynamic LDS per workgroup, so that a CU holds ONE
// workgroup of the launch (members must not share a CU: they would share its issue ports)
void sgx_trk3_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                     double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err,
                     int lds_pad) {
    if (lds_pad > 0)
        (void)hipFuncSetAttribute((const void*)trk3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_pad);
    trk3_kernel<<<n_blocks, T3_THREADS, (size_t)(lds_pad > 0 ? lds_pad : 0), st>>>(rec, codes, chans, out, done, K, prof, xch, err);
}
