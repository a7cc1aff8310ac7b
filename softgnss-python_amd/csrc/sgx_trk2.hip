// TrackingResult.track on gfx950, latency-mode kernel of round 2 (reference tracking.py:13-295; SURVEY.md section 9
// T1-T9).  Same decomposition as sgx_trk.hip - a channel's block is cut into units of 256 groups x 16 samples and
// P workgroups (one unit each) cooperate on a channel - but the per-block dependency chain
//     sums -> discriminators -> NCOs -> next block's parameters -> sums
// is rebuilt around what each link really depends on:
//
//   roles     a workgroup has 7 waves: 4 MAP waves (256 lanes, one 16-sample group each), a PLL wave, a DLL wave and a
//             RECORD wave.  One workgroup barrier per block hands the next block's parameters to the map waves.
//   map       the 16 bytes of a lane are loaded and converted to fp64 one block ahead (the next block's first sample
//             is known when the current block starts).  For each of the three code ramps the chip index at the group's
//             first sample and the switch sample follow from ONE fused evaluation t0 = ilo*step + start and the distance
//             to the next chip boundary in samples, u = (ceil(t0) - t0) / step: the reference's ramp
//             t(i) = fl(fl(i*step)+start) differs from the real one by < 2.3e-13 chips, so ceil(t(i)) equals the real
//             ramp's for every sample of the group unless a boundary lies within that distance of a sample - excluded
//             when frac(u) is outside [1e-7, 1 - 1e-7] samples (2.7e-9 chips).  A wave in which any lane fails the test
//             (probability ~1e-5 per block; all of block 0, whose prompt ramp starts ON a boundary) takes the exact
//             search of round 1 (ramp_setup).  Chip indices are therefore still bit-identical to
//             code[int64(ceil(linspace(...)))] (tracking.py:166-188); the chips themselves are two bits of a packed
//             sign table in LDS, read while the samples are accumulated.
//   reduce    six fp64 partials per lane -> transposing DPP reduction inside each row of 16 lanes (no LDS) -> 2^-32
//             fixed point -> integer LDS atomics (order-independent, hence deterministic) -> the wave that arrives last
//             publishes the member's six sums with ONE 64-bit integer atomic per sum into the channel's exchange line
//             in L2.  Every word carries an arrival count in its low 5 bits, lines are double-buffered by block parity
//             and never reset (consumers difference against the previous value), so there is nothing to zero and no
//             flag: the data is the flag.
//   filter    the PLL and DLL waves of EVERY member poll the line (one 16-byte / two 16-byte L1-bypassing loads), turn
//             the totals back into fp64 and run their half of the loop filter redundantly - bit-identical in all
//             members, nothing to broadcast - with short-chain arithmetic (sgx_trk_math.h): reciprocal-based division
//             and square root, a degree-8 Estrin atan on the locked range, Estrin sincos for the carrier tables, and
//             a division-free ceil for the block length that falls back to the IEEE division when the quotient is
//             within 4 ulp of an integer.  The end-of-block carrier phase and everything else that does not need the
//             sums is computed while the wave waits.
//   record    the 13 series values of a block are staged in LDS by member 0's filter waves and stored (to pinned host
//             memory, directly) by the record wave one block later, so no wave on the chain ever waits for a store.
//   abort     a poll that runs out of budget raises the channel's abort word; every member sees it in its next poll,
//             posts stop = 2 for the next block and the whole channel leaves the loop within one budget (the host then
//             repeats the launch with one workgroup per channel).
#include "sgx_trk_common.h"
#include "sgx_trk_math.h"

#define T2_MAP 256                 // map lanes = groups per unit
#define T2_THREADS 448             // 4 map waves + PLL wave (4) + DLL wave (5) + record wave (6)
#define T2_MAXP 16                 // members per channel (arrival tag: 5 bits)
#define T2_XCH_STRIDE 256          // 64-bit words per channel in the exchange area:
#define T2_XG 0                    //   [2 parities][6 words][16 members] granules
#define T2_XABORT 192              //   abort word
#define T2_XPLACE 208              //   [16] placement granules
#define T2_FIX 268435456.0         // 2^28: fixed-point scale of a granule's 48-bit payload (member sums are < 2^19)
#define T2_FIX16 524288.0          // 2^19: the same for two-byte samples (member sums are < 2^28)
// SB = bytes per IF sample (1: int8, 2: int16).  Positions are counted in samples everywhere; only the loads, the
// fixed-point scale and the reported file position (bytes, tracking.py:107 / fid.tell()) depend on it.
template <int SB>
__device__ __forceinline__ constexpr double t2_fix() { return SB == 1 ? T2_FIX : T2_FIX16; }
#define T2_MAGIC 6755399441055744.0   // 1.5 * 2^52: fl(x + MAGIC) holds round(x) in its low mantissa bits
#define T2_POLL_BUDGET (1 << 20)

// -DTRK_FINEPROF=1: time stamps at the natural synchronisation points only (poll exits, barriers) - undisturbed timing.
// -DTRK_FINEPROF=2: every probe, each preceded by a full wait - attributes the time inside a role, inflates the total.
#ifdef TRK_FINEPROF
#define T2STAMP(role, k)                                                                   \
    do {                                                                                   \
        if (role) {                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const long long t_ = (long long)__builtin_amdgcn_s_memtime();                 \
            fp[k] += t_ - fp_last;                                                         \
            fp_last = t_;                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)
#if TRK_FINEPROF >= 2
#define T2PROBE(role, k)                                                                   \
    do {                                                                                   \
        if (role) {                                                                        \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                    \
            T2STAMP(role, k);                                                              \
        }                                                                                  \
    } while (0)
#else
#define T2PROBE(role, k) do { } while (0)
#endif
#else
#define T2STAMP(role, k) do { } while (0)
#define T2PROBE(role, k) do { } while (0)
#endif

struct __attribute__((aligned(128))) T2Code {   // code side of a block's parameters (DLL wave -> everybody), by block parity
    // chain part: written right before the barrier that starts the block
    int blk;
    int stop;               // 1: the record ends inside this block (tracking.py:159-163); 2: a member gave up waiting;
                            // 3: the block does not fit the units of the launch
    int pad0[2];
    double step[3];         // ramp steps E, P, L (tracking.py:166-188)
    double inv_step;        // 1 / codePhaseStep (distances to chip boundaries in samples)
    // early part: known one block earlier (written while the previous block is processed)
    long long pos;          // record index of the block's first sample
    long long pad1;
    double start[4];        // ramp starts E, P, L
};

struct T2Carr {   // carrier side (PLL wave -> map waves), double-buffered: (cos, sin)(2 pi r m), r = turns per sample
    double2 T[52];    // [0..15]  B:  m = b         sample b of a group
                      // [16..31] W1: m = 16 a      group a = tid & 15
                      // [32..47] W2: m = 256 b     group row b = tid >> 4
                      // [48]     W3: first sample of this member's unit (4096 u - head), plus the block's start phase
};
#define T2_B 0
#define T2_W1 16
#define T2_W2 32
#define T2_W3 48

__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(ohi, olo);
}

template <int CTRL>
__device__ __forceinline__ long long dpp_movl(long long v) {
    const int lo = (int)(unsigned)(v & 0xFFFFFFFFll), hi = (int)(v >> 32);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return ((long long)ohi << 32) | (unsigned)olo;
}

// chip (as the high dword of +-1.0) of extended-code index k, k in [-1, 1026]
__device__ __forceinline__ unsigned chip_hi(const unsigned* s_chip, int k) { return s_chip[k + 1]; }

// Sign bits of the extended code in LDS: bit k + 1 of the packed table is set where chip k is -1 (k in [-1, 1054]).
// Two adjacent chips (k, k + 1) from one 8-byte read.
__device__ __forceinline__ unsigned chip_bits2(const unsigned* cbits, int k) {
    const int kk = (k < 0 ? 0 : (k > 1024 ? 1024 : k)) + 1;   // lanes beyond the block hold zeros: any chip will do
    const unsigned lo = cbits[kk >> 5], hi = cbits[(kk >> 5) + 1];
    const unsigned long long ww = ((unsigned long long)hi << 32) | lo;
    return (unsigned)(ww >> (kk & 31)) & 3u;
}

// chip index at sample ilo (k1) and the first sample with a larger index (isw) of the ramp t(i) = i*step + start, from
// one fused evaluation; `bad` is raised when a chip boundary lies within 1e-7 samples of a sample (then the exact
// search decides).
__device__ __forceinline__ void ramp_locate(double start, double step, double inv_step, double ilod, int ilo, int& k1,
                                            int& isw, bool& bad) {
    const double t0 = __builtin_fma(ilod, step, start);
    const double kd = ceil(t0);
    const double dist = kd - t0;                    // chips to the next boundary, in [0, 1)
    const double u = dist * inv_step;               // the same in samples (real arithmetic, ~1e-12)
    const double fu = floor(u);
    const double fr = u - fu;
    // no sample of the group within 1e-7 samples (2.7e-9 chips) of a boundary: the one ahead (fr) and, for the first
    // sample, the one just behind it (dist close to 1)
    bad = bad || !(fr > 1e-7 && fr < 1.0 - 1e-7 && dist < 1.0 - 3e-9);
    k1 = (int)kd;
    isw = ilo + (int)fu + 1;
}

// Carrier phasor tables of a block with rate w (rad/s), start phase rc and `head` bytes between the 16-byte boundary and
// the block's first sample (T5): one entry per lane, B | W1 | W2 | this member's W3 (lanes 48..63 all hold W3).
__device__ __forceinline__ void t2_carr_tables(double inv_2pifs_hi, double inv_2pifs_lo, double inv_2pi, double w, double rc,
                                               int head, int member, T2Carr& CN, int lane) {
    const double r_hi = w * inv_2pifs_hi;
    const double r_lo = __builtin_fma(w, inv_2pifs_hi, -r_hi) + w * inv_2pifs_lo;
    const int sel = lane >> 4, idx = lane & 15;
    const int mi = (sel == 3) ? (TRK_UNIT * member - head) : (idx << (4 * sel));
    const double mult = (double)mi;
    const double pp = r_hi * mult;
    const double ee = __builtin_fma(r_hi, mult, -pp) + r_lo * mult;
    double u = (pp - floor(pp)) + ee;
    const double u3 = u + rc * inv_2pi;        // < 2
    u = (sel == 3) ? (u3 - ((u3 >= 1.0) ? 1.0 : 0.0)) : u;
    double sn, cs;
    sgx_sincos_turns_short(u, sn, cs);
    // B, W1, W2 are consecutive 16-entry tables, W3 follows: lanes 48..63 all store the same W3
    CN.T[lane < 48 ? lane : 48] = make_double2(cs, sn);
}

// DLL-wave constants and the loop state it carries in registers.  Lanes work in parallel on the three ramps:
// lane & 3 = 0 early, 1 prompt, 2 late (3 repeats prompt); uniform results come from lane 1 / lane 0.
struct T2DllConst {
    double fs, inv_fs, code_len, spacing;
    double inv_nb_lane;     // RN(1 / (nb_base + (lane & 7))): reciprocals of the plausible block lengths, one per lane
    int nb_base;
    long long rec_len;
};

struct T2DllState {      // everything about the block being processed that the DLL wave needs again
    double rem;          // remCodePhase at the block's start
    long long pos;       // its first sample
    double step;         // codePhaseStep = codeFreq / fs (T1)
    double stp;          // the lane's ramp step (lane & 3: E, P, L, P)
    int blk;
};

// Chain part of the next block's parameters (T1, T3): block size and ramp steps from the new code frequency; the
// block starts at pos_n with code phase rem_n.  Writes N's chain part; returns the new state.
__device__ __forceinline__ T2DllState t2_code_chain(const T2DllConst& D, double codeFreq, double rem_n, long long pos_n,
                                                    bool gave_up, int P, T2Code& N, int lane) {
    const int l4 = lane & 3;
    const double off = (l4 == 0) ? -D.spacing : ((l4 == 2) ? D.spacing : 0.0);   // rem - spc == rem + (-spc) exactly
    const double step = div_rn(codeFreq, D.fs, D.inv_fs);                       // codeFreq / fs
    const int blk = sgx_ceil_div(D.code_len - rem_n, step);
    const double nb = (double)blk;
    const double span = nb * step;                                              // blksize * codePhaseStep
    const int ki = blk - D.nb_base;
    const bool known = (ki >= 0 && ki < 8);
    const int kq = __builtin_amdgcn_readfirstlane(ki) & 7;
    const double ynb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(D.inv_nb_lane), kq),
                                        __builtin_amdgcn_readlane(__double2loint(D.inv_nb_lane), kq));
    // np.linspace(start, stop, blk, endpoint=False): delta = stop - start; step = delta / blk
    const double start = rem_n + off;
    const double d = ((span + rem_n) + off) - start;
    double stp;
    if (__builtin_expect(known, 1)) stp = div_rn(d, nb, ynb);
    else stp = d / nb;
    if (lane < 3) N.step[lane] = stp;
    if (lane == 0) {
        const int stop = gave_up ? 2 : ((blk <= 0 || pos_n + blk > D.rec_len) ? 1 : ((blk + 15 > P * TRK_UNIT) ? 3 : 0));
        *reinterpret_cast<int4*>(&N.blk) = make_int4(blk, stop, 0, 0);
        N.inv_step = sgx_fast_rcp(step);
    }
    T2DllState st;
    st.rem = rem_n;
    st.pos = pos_n;
    st.step = step;
    st.stp = stp;
    st.blk = blk;
    return st;
}

// Code phase and first sample of the block after `st` (T4) and that block's ramp starts (its early part): nothing here
// needs the sums of `st`.
__device__ __forceinline__ void t2_code_late(const T2DllConst& D, const T2DllState& st, T2Code& N, int lane, double& rem_next,
                                             long long& pos_next) {
    const int l4 = lane & 3;
    const double off = (l4 == 0) ? -D.spacing : ((l4 == 2) ? D.spacing : 0.0);
    const double start = st.rem + off;
    const double t_last = ramp_at(st.blk - 1, st.stp, start);
    const double rn_lane = (t_last + st.step) - 1023.0;                         // T4 (meaningful in the prompt lane)
    rem_next = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rn_lane), 1),
                                __builtin_amdgcn_readlane(__double2loint(rn_lane), 1));
    pos_next = st.pos + st.blk;
    if (lane < 3) N.start[lane] = rem_next + off;
    if (lane == 0) N.pos = pos_next;
}

#define T2_PIN(x) asm volatile("" : "+v"(x))

// Everything a role needs that lives in LDS.
struct T2Shared {
    unsigned chip[1032];            // chip[k + 1] = chip of extended-code index k (tracking.py:111), exact-search path
    unsigned cbits[40];             // the same as packed sign bits: bit k + 1 set where chip k is -1
    T2Code code[2];
    T2Carr carr[2];
    double part[2][16][8];          // row sums of the map waves by block parity: [wave * 4 + row][word]
    unsigned ticket[2][2];          // arrival ticket of the map waves, by block parity
    double rec[2][16];              // a block's 13 series values (member 0), stored one block later
    int flag[4];                    // [0] same-XCD placement, [1] abort seen by this workgroup
    int rflag[4];                   // [0] PLL wave, [1] DLL wave: number of blocks whose record values are in rec[]
    long long tpub[2];              // (profiling) time stamp of the member's publish, by block parity
};

#ifdef TRK_FINEPROF
#define T2_FP_DECL long long fp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long fp_last = 0;
#define T2_FP_TOP fp_last = (long long)__builtin_amdgcn_s_memtime();
#define T2_FP_PRINT(role, lo, hi)                                                                                   \
    if (role) {                                                                                                     \
        for (int k = lo; k < hi; ++k) printf("[fineprof2] probe %2d: %8.1f cycles/block\n", k, (double)fp[k] / ms); \
    }
#else
#define T2_FP_DECL
#define T2_FP_TOP
#define T2_FP_PRINT(role, lo, hi)
#endif

// a lane's 16 samples as loaded: one 16-byte word of int8, or two of int16
template <int SB> struct T2Raw;
template <> struct T2Raw<1> { uint4 a; };
template <> struct T2Raw<2> { uint4 a, b; };

template <int SB>
__device__ __forceinline__ T2Raw<SB> t2_load(const int8_t* __restrict__ rec, long long first_sample, long long limit) {
    T2Raw<SB> r;
    if constexpr (SB == 1) {
        r.a = load_group(rec, first_sample, limit);
    } else {
        r.a = load_group(rec, first_sample * 2, limit);
        r.b = load_group(rec, first_sample * 2 + 16, limit);
    }
    return r;
}

// 16 int16 samples -> fp64, samples before the block's first one (i0 + b < 0) zeroed
__device__ __forceinline__ void t2_convert(const T2Raw<2>& raw, int i0, double (&xd)[16]) {
#define T2_CV(b, w, hi)                                                                          \
    {                                                                                            \
        const int xi_ = hi ? ((int)(w) >> 16) : (int)(short)((w) & 0xFFFF);                      \
        xd[b] = (i0 + b >= 0) ? (double)xi_ : 0.0;                                               \
    }
    T2_CV(0, raw.a.x, 0) T2_CV(1, raw.a.x, 1) T2_CV(2, raw.a.y, 0) T2_CV(3, raw.a.y, 1)
    T2_CV(4, raw.a.z, 0) T2_CV(5, raw.a.z, 1) T2_CV(6, raw.a.w, 0) T2_CV(7, raw.a.w, 1)
    T2_CV(8, raw.b.x, 0) T2_CV(9, raw.b.x, 1) T2_CV(10, raw.b.y, 0) T2_CV(11, raw.b.y, 1)
    T2_CV(12, raw.b.z, 0) T2_CV(13, raw.b.z, 1) T2_CV(14, raw.b.w, 0) T2_CV(15, raw.b.w, 1)
#undef T2_CV
}

// 16 int8 samples -> fp64, samples before the block's first one (i0 + b < 0) zeroed
__device__ __forceinline__ void t2_convert(const T2Raw<1>& raw1, int i0, double (&xd)[16]) {
    const uint4& raw = raw1.a;
#define T2_CV(b, w, sh)                                                                          \
    {                                                                                            \
        const int xi_ = (sh == 24) ? ((int)(w) >> 24) : (int)(signed char)(((w) >> sh) & 0xFF);  \
        xd[b] = (i0 + b >= 0) ? (double)xi_ : 0.0;                                               \
    }
    T2_CV(0, raw.x, 0) T2_CV(1, raw.x, 8) T2_CV(2, raw.x, 16) T2_CV(3, raw.x, 24)
    T2_CV(4, raw.y, 0) T2_CV(5, raw.y, 8) T2_CV(6, raw.y, 16) T2_CV(7, raw.y, 24)
    T2_CV(8, raw.z, 0) T2_CV(9, raw.z, 8) T2_CV(10, raw.z, 16) T2_CV(11, raw.z, 24)
    T2_CV(12, raw.w, 0) T2_CV(13, raw.w, 8) T2_CV(14, raw.w, 16) T2_CV(15, raw.w, 24)
#undef T2_CV
}

// ================================ MAP (waves 0-3) ================================
template <int SB>
__device__ __forceinline__ int t2_map_role(T2Shared& S, const int8_t* __restrict__ rec, long long rec_alloc, int ms,
                                           long long pos0, int member, int tid, unsigned long long* __restrict__ xbase,
                                           bool fast, bool prof_on, bool prof_any) {
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long long limit = rec_alloc - 16;                  // bytes: the last 16-byte word that may be loaded
    const int g = (tid & 255) + member * T2_MAP;             // the lane's group inside the block's aligned window
    const long long lane_off = (long long)g * 16;
    T2_FP_DECL
    (void)prof_on;
    // state prepared one block ahead: the lane's 16 samples as fp64 and where they sit in the block
    double xd[16];
    int i0, ilo;
    double ilod;
    T2Raw<SB> raw;
    int blk_pred;                    // block length the prepared samples are cut for

    // samples from block sample index END on belong to the next block: zeroed by an integer mask on the high dword
#define T2_CUT(END)                                                                                            \
    do {                                                                                                       \
        if (i0 < (END) && i0 + 16 > (END)) {                                                                   \
            const int e_ = (END) - i0;                                                                         \
            _Pragma("unroll") for (int b_ = 0; b_ < 16; ++b_)                                                  \
                xd[b_] = __hiloint2double(__double2hiint(xd[b_]) & ((b_ - e_) >> 31), 0);                      \
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
        t2_convert(raw, i0, xd);                                                                               \
        blk_pred = (BLK_PRED);                                                                                 \
        T2_CUT(blk_pred);                                                                                      \
    } while (0)

    raw = t2_load<SB>(rec, (pos0 & ~15ll) + lane_off, limit);
    T2_PREPARE(pos0, S.code[0].blk);
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        const T2Code& C = S.code[par];
        // one batch of LDS reads: chain part, early part, the lane's carrier phasors
        const int4 hd = *reinterpret_cast<const int4*>(&C.blk);      // blk, stop
        const double stepE = C.step[0], stepP = C.step[1], stepL = C.step[2], inv_step = C.inv_step;
        const double startE = C.start[0], startP = C.start[1], startL = C.start[2];
        const long long pos = C.pos;
        const T2Carr& CR = S.carr[par];
        const double2 w1 = CR.T[T2_W1 + (tid & 15)], w2 = CR.T[T2_W2 + ((tid >> 4) & 15)], w3 = CR.T[T2_W3];
        if (hd.y) break;
        T2_FP_TOP
        __builtin_amdgcn_s_setprio(2);
        const int blk = hd.x;
        const long long pos_next = pos + blk;
        const T2Raw<SB> nraw = t2_load<SB>(rec, (pos_next & ~15ll) + lane_off, limit);   // next block's bytes
        T2PROBE(prof_on, 0);   // parameters read, next block's load issued
        int kE, kP, kL, swE, swP, swL;
        bool bad = false;
        ramp_locate(startE, stepE, inv_step, ilod, ilo, kE, swE, bad);
        ramp_locate(startP, stepP, inv_step, ilod, ilo, kP, swP, bad);
        ramp_locate(startL, stepL, inv_step, ilod, ilo, kL, swL, bad);
        if (__builtin_expect(__any(bad && i0 < blk), 0)) {
            // a chip boundary within 1e-7 samples of a sample somewhere in this wave: exact search (round-1 path)
            ramp_setup(startE, stepE, inv_step, ilo, kE, swE);
            ramp_setup(startP, stepP, inv_step, ilo, kP, swP);
            ramp_setup(startL, stepL, inv_step, ilo, kL, swL);
        }
        // chips k1 and k1 + 1 of every ramp (two sign bits each); they are needed only after the accumulation
        const unsigned bE = chip_bits2(S.cbits, kE), bP = chip_bits2(S.cbits, kP), bL = chip_bits2(S.cbits, kL);
        // group-start phasor G = W1[tid & 15] * W2[(tid >> 4) & 15] * W3; a group that lies entirely beyond the block
        // (its samples belong to the next one) gets a zero phasor, i.e. adds nothing
        double gc, gs;
        {
            const double lc = __builtin_fma(w1.x, w2.x, -(w1.y * w2.y));
            const double ls = __builtin_fma(w1.x, w2.y, w1.y * w2.x);
            gc = __builtin_fma(lc, w3.x, -(ls * w3.y));
            gs = __builtin_fma(lc, w3.y, ls * w3.x);
            const bool beyond = i0 >= blk;
            gc = beyond ? 0.0 : gc;
            gs = beyond ? 0.0 : gs;
        }
        if (__builtin_expect(blk != blk_pred, 0)) {
            // the block is a sample longer or shorter than predicted (about one block in ten): the lanes around its
            // end convert their bytes again and cut them at the real length
            const int lo_ = blk < blk_pred ? blk : blk_pred, hi_ = blk < blk_pred ? blk_pred : blk;
            if (__any(i0 < hi_ && i0 + 16 > lo_)) {
                t2_convert(raw, i0, xd);
                T2_CUT(blk);
            }
        }
        T2PROBE(prof_on, 1);   // switch samples resolved
        const double cE1 = __hiloint2double((int)(0x3FF00000u | (bE << 31)), 0), cE2 = __hiloint2double((int)(0x3FF00000u | ((bE >> 1) << 31)), 0);
        const double cP1 = __hiloint2double((int)(0x3FF00000u | (bP << 31)), 0), cP2 = __hiloint2double((int)(0x3FF00000u | ((bP >> 1) << 31)), 0);
        const double cL1 = __hiloint2double((int)(0x3FF00000u | (bL << 31)), 0), cL2 = __hiloint2double((int)(0x3FF00000u | ((bL >> 1) << 31)), 0);
        double aIE, aQE, aIP, aQP, aIL, aQL;
        const int iend = i0 + 16;
        int swmin = swE < swP ? swE : swP;
        swmin = swL < swmin ? swL : swmin;
        const bool eS = (swE == swmin), pS = (swP == swmin), lS = (swL == swmin);
        const bool odd = (swE < iend && !eS) || (swP < iend && !pS) || (swL < iend && !lS);
        if (__builtin_expect(__any(odd), 0)) {
            // exact per-sample path (a ramp switches at a second position inside the group)
            aIE = aQE = aIP = aQP = aIL = aQL = 0.0;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int i = i0 + b;
                const double2 Bb = CR.T[T2_B + b];
                const double c = __builtin_fma(gc, Bb.x, -(gs * Bb.y));
                const double s = __builtin_fma(gs, Bb.x, gc * Bb.y);
                const double xs = s * xd[b], xc = c * xd[b];
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
            // samples b >= bsw come after the switch.  The samples are small integers, so their fp64 low dword is zero
            // and masking the HIGH dword alone zeroes one: xt_hi = xd_hi & ((b - bsw) >> 31 ? 0 : ~0) - plain integer
            // VALU, no compare/select round trip through VCC per sample
            int bsw = swmin - i0;
            bsw = bsw > 16 ? 16 : bsw;
            double Ac = 0.0, As = 0.0, Tc = 0.0, Ts = 0.0;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const double2 Bb = CR.T[T2_B + b];
                Ac = __builtin_fma(xd[b], Bb.x, Ac);
                As = __builtin_fma(xd[b], Bb.y, As);
                const int keep = ~((b - bsw) >> 31);                  // all ones iff b >= bsw
                const double xt = __hiloint2double(__double2hiint(xd[b]) & keep, 0);
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
            aIE = __builtin_fma(dE, tlI, cE1 * allI);
            aQE = __builtin_fma(dE, tlQ, cE1 * allQ);
            aIP = __builtin_fma(dP, tlI, cP1 * allI);
            aQP = __builtin_fma(dP, tlQ, cP1 * allQ);
            aIL = __builtin_fma(dL, tlI, cL1 * allI);
            aQL = __builtin_fma(dL, tlQ, cL1 * allQ);
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
                // {16-bit epoch tag | 48-bit two's-complement fixed point}, ONE aligned 8-byte store
                const double t = __builtin_fma(v, t2_fix<SB>(), T2_MAGIC);
                const unsigned long long q = (unsigned long long)(__double_as_longlong(t) - __double_as_longlong(T2_MAGIC));
                const unsigned long long gran = ((unsigned long long)((unsigned)(it + 1) & 0xFFFFu) << 48) | (q & 0xFFFFFFFFFFFFull);
                granule_store(xbase + T2_XG + par * 96 + word * 16 + member, gran, fast);
            }
            if (prof_any && lane == 0) S.tpub[par] = (long long)__builtin_amdgcn_s_memtime();
        }
        T2STAMP(prof_on, 5);   // published (or handed to the wave that publishes)
        // ---- shadow: prepare the next block with this block's rates ----
        __builtin_amdgcn_s_setprio(0);
        raw = nraw;
        T2_PREPARE(pos_next, blk);
        T2STAMP(prof_on, 6);   // next block prepared
        wg_barrier();
        T2STAMP(prof_on, 7);   // waiting for the loop filter
    }
    T2_FP_PRINT(prof_on && lane == 0, 0, 8)
    return it;
}

// ================================ PLL (wave 4) ================================
template <int SB>
__device__ __forceinline__ int t2_pll_role(T2Shared& S, const TrkConst& K, const TrkChan& cc, int member, int lane, int P,
                                           int ch, unsigned long long* __restrict__ xbase, int* __restrict__ err,
                                           bool prof_on, long long* __restrict__ prof) {
    // tracking.py:123-130
    long long acc_map = 0, acc_xch = 0, acc_flt = 0, t_top = 0, t_arr = 0;   // SGX_TRK_PROFILE=1: per-member phase times
    double carrBasis = cc.acquiredFreq;
    double remCarr = 0.0, w_cur = (cc.acquiredFreq * 2.0) * M_PI, oldCarrNco = 0.0, oldCarrErr = 0.0;
    const double two_pi = 2 * M_PI;
    // constants of the call in registers (kernel arguments would be re-fetched through the scalar cache on the chain)
    double k_a = K.k_carr_a, k_b = K.k_carr_b, inv_pi = K.inv_pi, inv_2pi = K.inv_2pi, c_hi = K.inv_2pifs_hi,
           c_lo = K.inv_2pifs_lo, inv_fs = K.inv_fs, fs = K.fs;
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(inv_pi); T2_PIN(inv_2pi); T2_PIN(c_hi); T2_PIN(c_lo); T2_PIN(inv_fs); T2_PIN(carrBasis);
    T2_PIN(fs);
    const int ms = K.ms;
    // lane = 16 word + member polls that member's granule of I_P (word 0, row 0) / Q_P (word 1, row 1)
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
        if (member == 0 && it > 0) {
            if (lane == 0) {
                double* R = S.rec[par ^ 1];      // T9 record (tracking.py:255-275) of block it - 1, stored by the record wave
                R[2] = r_cf;
                R[3] = r_ip;
                R[7] = r_qp;
                R[11] = r_err;
                R[12] = r_nco;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) *(volatile int*)&S.rflag[0] = it;
        }
        if (hd.y) break;
        T2_FP_TOP
        if (prof) t_top = (long long)__builtin_amdgcn_s_memtime();
        // before the sums arrive: carrier phase at the end of this block (T5), exact remainder by FMA
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
        T2_PIN(rc);   // (keeps the block-end phase computation ahead of the wait)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long* gp = xbase + T2_XG + par * 96 + (lane & 31);
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
        for (;;) {
            if (mine) x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(!mine || (x >> 48) == tag)) break;
            if ((--budget & 31) == 0) {
                if (budget == 0 || *(volatile int*)&S.flag[1] != 0) {
                    gave_up = true;
                    break;
                }
            }
        }
        T2STAMP(prof_on, 8);   // waiting for the sums
        if (prof) {
            t_arr = (long long)__builtin_amdgcn_s_memtime();
            const long long tp = *(volatile long long*)&S.tpub[par];
            acc_map += tp - t_top;       // barrier release -> this member's publish
            acc_xch += t_arr - tp;       // this member's publish -> every member's sums visible
        }
        // sum of the members' payloads (integers: exact, order-free), rows of 16 lanes
        long long q = mine ? ((long long)(x << 16) >> 16) : 0ll;
        q += dpp_movl<0xB1>(q);
        q += dpp_movl<0x4E>(q);
        q += dpp_movl<0x141>(q);
        q += dpp_movl<0x140>(q);
        const double v = (double)q * (1.0 / t2_fix<SB>());
        const double I_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                            __builtin_amdgcn_readlane(__double2loint(v), 0));
        const double Q_P = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16),
                                            __builtin_amdgcn_readlane(__double2loint(v), 16));
        // T7 PLL (tracking.py:223-235)
        const double carrError = sgx_div_with_rcp(sgx_atan_ratio(Q_P, I_P) * 0.5, M_PI, inv_pi);   // atan(Q/I) / 2 / pi
        const double carrNco = oldCarrNco + k_a * (carrError - oldCarrErr) + carrError * k_b;
        const double carrFreq = carrBasis + carrNco;
        const double w_new = (carrFreq * 2.0) * M_PI;
        oldCarrNco = carrNco;
        oldCarrErr = carrError;
        T2PROBE(prof_on, 9);   // discriminator + NCO
        // carrier tables of the next block
        if (it + 1 < ms) t2_carr_tables(c_hi, c_lo, inv_2pi, w_new, rc, head_next, member, S.carr[par ^ 1], lane);
        w_cur = w_new;
        remCarr = rc;
        r_cf = carrFreq;
        r_ip = I_P;
        r_qp = Q_P;
        r_err = carrError;
        r_nco = carrNco;
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            atomicExch(err, 1 + ch);
            __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        T2STAMP(prof_on, 10);  // carrier tables
        wg_barrier();
        __builtin_amdgcn_s_setprio(0);   // what follows until the next wait is off the chain: let the map waves issue first
        if (prof) acc_flt += (long long)__builtin_amdgcn_s_memtime() - t_arr;   // sums visible -> barrier released (both filter waves done)
        T2STAMP(prof_on, 11);
    }
    if (member == 0 && it > 0 && it == ms) {
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
        if (lane == 0) *(volatile int*)&S.rflag[0] = it;
    }
    if (prof && lane == 0) {
        prof[ch * 64 + member] = acc_map;
        prof[ch * 64 + 16 + member] = acc_xch;
        prof[ch * 64 + 32 + member] = acc_flt;
    }
    T2_FP_PRINT(prof_on && lane == 0, 8, 12)
    return it;
}

// ================================ DLL (wave 5) ================================
template <int SB>
__device__ __forceinline__ int t2_dll_role(T2Shared& S, const TrkConst& K, const T2DllConst& D, T2DllState st, int member,
                                           int lane, int P, int ch, unsigned long long* __restrict__ xbase,
                                           int* __restrict__ err, bool prof_on, long long file_off) {
    // tracking.py:114-121; `st` describes block 0 (its chain and early parts are posted)
    double oldCodeNco = 0.0, oldCodeErr = 0.0;
    double k_a = K.k_code_a, k_b = K.k_code_b, basis = K.code_basis;
    T2_PIN(k_a); T2_PIN(k_b); T2_PIN(basis);
    const int ms = K.ms;
    // lane = 16 row + member polls that member's granule of word 2 + row: rows I_E, Q_E, I_L, Q_L
    const int l4 = lane & 3;
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
        const int stop = C.stop;
        if (member == 0 && it > 0) {
            double* R = S.rec[par ^ 1];
            // I_E -> 4, Q_E -> 6, I_L -> 5, Q_L -> 8 (series order of _native.SERIES)
            if (lane < 4) R[lane == 0 ? 4 : (lane == 1 ? 6 : (lane == 2 ? 5 : 8))] = r_v;
            if (lane == 0) {
                R[0] = (double)(st.pos * SB + file_off);   // position after block it - 1 = first sample of block it
                R[1] = r_cf;
                R[9] = r_err;
                R[10] = r_nco;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) *(volatile int*)&S.rflag[1] = it;
        }
        if (stop) break;
        T2_FP_TOP
        // late part of this block and early part of the next (nothing here needs the sums)
        double rem_next;
        long long pos_next;
        t2_code_late(D, st, S.code[par ^ 1], lane, rem_next, pos_next);
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long* gp = xbase + T2_XG + par * 96 + 32 + lane;
        const unsigned long long tag = (unsigned long long)((unsigned)(it + 1) & 0xFFFFu);
        unsigned long long x = 0, xa = 0;
        int budget = T2_POLL_BUDGET;
        bool gave_up = false;
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
        T2STAMP(prof_on, 12);  // waiting for the sums
        // T8 DLL (tracking.py:238-251).  Row r of the wave holds the members' payloads of I_E | Q_E | I_L | Q_L: integer
        // row sums (exact, order-free), then lanes 0..3 of every quad take the four totals: v = I_E | Q_E | I_L | Q_L
        long long q = mine ? ((long long)(x << 16) >> 16) : 0ll;
        q += dpp_movl<0xB1>(q);
        q += dpp_movl<0x4E>(q);
        q += dpp_movl<0x141>(q);
        q += dpp_movl<0x140>(q);
        const double vr = (double)q * (1.0 / t2_fix<SB>());
        const int vh = __double2hiint(vr), vl = __double2loint(vr);
        const int h0 = __builtin_amdgcn_readlane(vh, 0), l0 = __builtin_amdgcn_readlane(vl, 0);
        const int h1 = __builtin_amdgcn_readlane(vh, 16), l1 = __builtin_amdgcn_readlane(vl, 16);
        const int h2 = __builtin_amdgcn_readlane(vh, 32), l2 = __builtin_amdgcn_readlane(vl, 32);
        const int h3 = __builtin_amdgcn_readlane(vh, 48), l3 = __builtin_amdgcn_readlane(vl, 48);
        const double v = __hiloint2double(l4 == 0 ? h0 : (l4 == 1 ? h1 : (l4 == 2 ? h2 : h3)),
                                          l4 == 0 ? l0 : (l4 == 1 ? l1 : (l4 == 2 ? l2 : l3)));
        const double sq = v * v;
        const double e2 = sq + dpp_mov<0xB1>(sq);            // lanes 0,1: I_E^2 + Q_E^2; lanes 2,3: I_L^2 + Q_L^2
        const double mag = sgx_fast_sqrt(e2);                // E | E | L | L
        const double oth = dpp_mov<0x4E>(mag);               // L | L | E | E
        const double ce_lane = sgx_fast_div(mag - oth, mag + oth);   // lanes 0,1: (E - L) / (E + L)
        const double codeError = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ce_lane), 0),
                                                  __builtin_amdgcn_readlane(__double2loint(ce_lane), 0));
        const double codeNco = oldCodeNco + k_a * (codeError - oldCodeErr) + codeError * k_b;
        const double cf_new = basis - codeNco;
        oldCodeNco = codeNco;
        oldCodeErr = codeError;
        T2PROBE(prof_on, 13);  // discriminator + NCO
        // T1, T3: block size and ramp steps of the next block
        const T2DllState nx = t2_code_chain(D, cf_new, rem_next, pos_next, gave_up, P, S.code[par ^ 1], lane);
        if (lane == 0 && nx.blk + 15 > P * TRK_UNIT && !gave_up) atomicExch(err, TRK_ERR_RANGE | (1 + ch));
        st = nx;
        r_v = v;
        r_cf = cf_new;
        r_err = codeError;
        r_nco = codeNco;
        if (gave_up && lane == 0) {
            S.flag[1] = 1;
            if (xa == 0) {
                atomicExch(err, 1 + ch);
                __hip_atomic_store(xabort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        T2STAMP(prof_on, 14);  // next block's code parameters
        wg_barrier();
        __builtin_amdgcn_s_setprio(0);
        T2STAMP(prof_on, 15);
    }
    if (member == 0 && it > 0 && it == ms) {
        double* R = S.rec[(it - 1) & 1];
        if (lane < 4) R[lane == 0 ? 4 : (lane == 1 ? 6 : (lane == 2 ? 5 : 8))] = r_v;
        if (lane == 0) {
            R[0] = (double)(st.pos * SB + file_off);
            R[1] = r_cf;
            R[9] = r_err;
            R[10] = r_nco;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) *(volatile int*)&S.rflag[1] = it;
    }
    T2_FP_PRINT(prof_on && lane == 0, 12, 16)
    return it;
}

// ================================ RECORD (wave 6) ================================
// Stores block k's 13 series values once both filter waves have posted them (rflag >= k + 1): one block behind.
__device__ __forceinline__ void t2_rec_store(T2Shared& S, int k, long long m, int lane, double* __restrict__ o) {
    int budget = 1 << 16;
    while ((*(volatile int*)&S.rflag[0] < k + 1 || *(volatile int*)&S.rflag[1] < k + 1) && --budget) __builtin_amdgcn_s_sleep(2);
    if (lane < SGX_NUM_SERIES) o[lane * m + k] = S.rec[k & 1][lane];
}

__device__ __forceinline__ int t2_rec_role(T2Shared& S, int ms, int member, int lane, double* __restrict__ o) {
    const long long m = ms;
    int it = 0;
    for (; it < ms; ++it) {
        const int par = it & 1;
        if (S.code[par].stop) break;
        if (member == 0 && it > 0) t2_rec_store(S, it - 1, m, lane, o);
        wg_barrier();
    }
    return it;
}

template <int SB>
__global__ __launch_bounds__(T2_THREADS) void trk2_kernel(const int8_t* __restrict__ rec, const int8_t* __restrict__ codes,
                                                          const TrkChan* __restrict__ chans, double* __restrict__ out,
                                                          int* __restrict__ ms_done, TrkConst K,
                                                          long long* __restrict__ prof,
                                                          unsigned long long* __restrict__ xch, int* __restrict__ err) {
    __shared__ T2Shared S;
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
    unsigned long long* __restrict__ xbase = xch + (long long)ch * T2_XCH_STRIDE;   // granules, abort word, placement granules
    unsigned long long* const xabort = xbase + T2_XABORT;
    const bool prof_on = (member == 0 && ch == 0 && prof != nullptr && (wave == 0 || wave == 4 || wave == 5));

    // ---- placement: are all members of the channel on one XCD (one L2)?  Then the exchange may stay in that L2.
    if (tid < 4) {
        S.flag[tid] = 0;
        S.rflag[tid] = 0;
        S.ticket[tid >> 1][tid & 1] = 0;
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
                atomicExch(err, 1 + ch);
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

    // block 0 parameters (tracking.py:114-130): chain part and early part
    T2DllConst D;
    T2DllState st0;
    st0.rem = 0.0;
    st0.pos = cc.pos0;
    st0.step = 0.0;
    st0.stp = 0.0;
    st0.blk = 0;
    if (wave == 5) {
        D.fs = K.fs;
        D.inv_fs = K.inv_fs;
        D.code_len = K.code_len;
        D.spacing = K.spacing;
        D.inv_nb_lane = 1.0 / (double)(K.nb_base + (lane & 7));
        D.nb_base = K.nb_base;
        D.rec_len = (K.rec_len - cc.pad) / SB;      // samples on the channel's grid (cc.pad: its byte shift, SB = 2 only)
        st0 = t2_code_chain(D, K.code_basis, 0.0, cc.pos0, false, P, S.code[0], lane);
        const double off = ((lane & 3) == 0) ? -K.spacing : (((lane & 3) == 2) ? K.spacing : 0.0);
        if (lane < 3) S.code[0].start[lane] = 0.0 + off;
        if (lane == 0) {
            S.code[0].pos = cc.pos0;
            if (st0.blk + 15 > P * TRK_UNIT) atomicExch(err, TRK_ERR_RANGE | (1 + ch));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0 && dead) S.code[0].stop = 2;
    }
    if (wave == 4)
        t2_carr_tables(K.inv_2pifs_hi, K.inv_2pifs_lo, K.inv_2pi, (cc.acquiredFreq * 2.0) * M_PI, 0.0, (int)(cc.pos0 & 15),
                       member, S.carr[0], lane);
    __syncthreads();

    double* __restrict__ o = out + (long long)ch * SGX_NUM_SERIES * K.ms;
    int done;
    if (wave < 4) done = t2_map_role<SB>(S, rec + cc.pad, K.rec_alloc, K.ms, cc.pos0, member, tid, xbase, fast, prof_on, prof != nullptr);
    else if (wave == 4) done = t2_pll_role<SB>(S, K, cc, member, lane, P, ch, xbase, err, prof_on, prof);
    else if (wave == 5) done = t2_dll_role<SB>(S, K, D, st0, member, lane, P, ch, xbase, err, prof_on, K.file_off + cc.pad);
    else done = t2_rec_role(S, K.ms, member, lane, o);

    // a channel that was given up reports the blocks completed before the abort
    const bool aborted = S.code[done & 1].stop == 2;
    if (wave == 6 && member == 0 && done > 0 && !aborted) t2_rec_store(S, done - 1, (long long)K.ms, lane, o);
    if (aborted && done > 0) done -= 1;
    if (tid == 0 && member == 0) ms_done[ch] = done;
}

void sgx_trk2_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                     double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err,
                     int sample_bytes) {
    if (sample_bytes == 2) trk2_kernel<2><<<n_blocks, T2_THREADS, 0, st>>>(rec, codes, chans, out, done, K, prof, xch, err);
    else trk2_kernel<1><<<n_blocks, T2_THREADS, 0, st>>>(rec, codes, chans, out, done, K, prof, xch, err);
}
