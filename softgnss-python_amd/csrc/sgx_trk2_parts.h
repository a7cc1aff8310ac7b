// Device helpers of the latency-mode tracking kernels (sgx_trk2.hip: one workgroup per unit / per unit and correlator
// arm; sgx_trk3.hip: the speculative kernel): exchange granules, per-block parameter records in LDS, 64-bit integer DPP
// reductions, chip sign bits, the fused ramp evaluation, carrier table entries, LDS flags, the 16-sample loads.
#pragma once
#include "sgx_trk_common.h"
#include "sgx_trk_math.h"

#define T2_MAP 256                 // map lanes = groups per unit
#define T2_THREADS 448             // 4 map waves + PLL wave (4) + DLL wave (5) + record wave (6)
#define T2_MAXP 16                 // units per channel
#define T2_MAXM 48                 // members per channel (3 arms x 16 units)
// The exchange area of a channel, in 64-bit words: 12 granule lines [2 parities][6 sums] of 16 units each, T2_XLINE words
// apart, then the abort word and 48 placement granules.  (sgx_trk.hip sizes the allocation with the same T2_XCH_STRIDE.)
#ifndef T2_XLINE
#define T2_XLINE 16                // 128 bytes: the lines are adjacent
#endif
#define T2_XG 0
#define T2_XABORT (12 * T2_XLINE)
#define T2_XPLACE (12 * T2_XLINE + 8)
#define T2_XCH_STRIDE (((12 * T2_XLINE + 8 + 48) + 255) / 256 * 256)
#define T2_PROF_STRIDE 192         // profile words per channel: [3 phases][64 members]
#define T2_FIX 268435456.0         // 2^28: fixed-point scale of a granule's 48-bit payload (member sums are < 2^19)
#define T2_FIX16 524288.0          // 2^19: the same for two-byte samples (member sums are < 2^28)
// SB = bytes per IF sample (1: int8 / uint8, 2: int16, 4: float32, 8: float64).  Positions are counted in samples
// everywhere; only the loads, the fixed-point scale and the reported file position (bytes, tracking.py:107 / fid.tell(),
// tracking.py:255) depend on it.  Float samples are scaled on conversion by a power of two of the record's (TrkConst::fscale,
// exact) so that none exceeds 128: they then share the int8 scale of the granules, and the host scales the correlator
// series back (exact again).
template <int SB>
__device__ __forceinline__ constexpr double t2_fix() { return SB == 2 ? T2_FIX16 : T2_FIX; }
// ... when a member owns ceil(n_units / P) units its sum can be that many times larger: the scale drops by the next
// power of two (exact), so the 48-bit payload still holds it
template <int SB>
__device__ __forceinline__ double t2_fix_of(int P, int n_units, bool uns) {
    const int u = (n_units + P - 1) / P;
    const int sh = ((u <= 1) ? 0 : (32 - __builtin_clz((unsigned)(u - 1)))) + (uns ? 1 : 0);   // (bytes up to 255: one bit)
    return __hiloint2double(__double2hiint(t2_fix<SB>()) - (sh << 20), 0);
}
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
    double inv_step;        // ~1 / step (2^-48): distances to chip boundaries in samples
    double step;            // codeFreq / fs to 3 ulp: the slope of the real ramps (the guard covers the difference)
    // early part: known one block earlier (written while the previous block is processed)
    double start_arm;       // ramp start of this workgroup's arm (ARMS = 1; shares a 16-byte read with step)
    double start[3];        // ramp starts E, P, L
    long long pos;          // record index of the block's first sample
    // exact part: posted right after the barrier that starts the block (the exact search needs it, ~1e-5 of the waves)
    double stp[3];          // linspace steps E, P, L (tracking.py:166-188)
    int xflag;              // block number + 1 once stp[] is valid (lds_peek / lds_poke)
    int pad;
};

struct T2Carr {   // carrier side (PLL wave -> map waves), double-buffered: (cos, sin)(2 pi r m), r = turns per sample
    double2 T[64];    // [0..15]  B:  m = b         sample b of a group
                      // [16..31] W1: m = 16 a      group a = tid & 15
                      // [32..47] W2: m = 256 b     group row b = tid >> 4
                      // [48 + j] W3: first sample of this member's j-th unit (4096 (u + j P) - head), plus the block's
                      //              start phase
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

// row_bcast:15 (lane 15 of every row to the next row) / row_bcast:31 (lane 31 to rows 2 and 3); other rows get 0
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_bcast(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xF, false);
    return __hiloint2double(ohi, olo);
}

template <int CTRL>
__device__ __forceinline__ long long dpp_movl(long long v) {
    const int lo = (int)(unsigned)(v & 0xFFFFFFFFll), hi = (int)(v >> 32);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return ((long long)ohi << 32) | (unsigned)olo;
}

// b + dpp(a) on 64-bit integers with the DPP source fused into the two adds (VOP2 DPP forms: two instructions per
// step instead of two moves and two adds).  s_nop 1: a DPP source written by the previous VALU instruction needs two
// wait states, which the assembler does not insert inside an asm block.
#define T2_DPP_ADDL(NAME, CTRLSTR)                                                                                  \
    __device__ __forceinline__ unsigned long long NAME(unsigned long long a, unsigned long long b) {               \
        const unsigned alo = (unsigned)a, ahi = (unsigned)(a >> 32), blo = (unsigned)b, bhi = (unsigned)(b >> 32);  \
        unsigned olo, ohi;                                                                                          \
        asm volatile("s_nop 1\n\t"                                                                                  \
                     "v_add_co_u32_dpp %0, vcc, %2, %4 " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"     \
                     "v_addc_co_u32_dpp %1, vcc, %3, %5, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1"   \
                     : "=&v"(olo), "=&v"(ohi)                                                                       \
                     : "v"(alo), "v"(ahi), "v"(blo), "v"(bhi)                                                       \
                     : "vcc");                                                                                      \
        return ((unsigned long long)ohi << 32) | olo;                                                               \
    }
T2_DPP_ADDL(dpp_addl_xor1, "quad_perm:[1,0,3,2]")
T2_DPP_ADDL(dpp_addl_xor2, "quad_perm:[2,3,0,1]")
T2_DPP_ADDL(dpp_addl_ror4, "row_ror:4")
T2_DPP_ADDL(dpp_addl_ror8, "row_ror:8")
T2_DPP_ADDL(dpp_addl_hmir, "row_half_mirror")
T2_DPP_ADDL(dpp_addl_mir, "row_mirror")

// Sum of the 48-bit payloads (two's complement) of a row's sixteen granules, as a double in every lane of the row: the payload
// as a low limb of 24 bits and a sign-extended high limb, two INDEPENDENT chains of one DPP add per step - no add-with-carry
// (an instruction that waits for VCC from the one before it costs a lone wave 10-20 cycles; round 5, sgx_trk3.hip).
__device__ __forceinline__ double t2_sum48_row(unsigned long long x) {
    const unsigned xl = (unsigned)x, xh = (unsigned)(x >> 32);
    unsigned lo = xl & 0xFFFFFFu;
    unsigned hi = (unsigned)((int)(__builtin_amdgcn_alignbit(xh, xl, 24) << 8) >> 8);   // bits 24..47, sign-extended
    unsigned a, b;
#define T2_S48(d0, d1, s0, s1, ctl)                                                               \
        "v_add_u32_dpp " d0 ", " s0 ", " s0 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32_dpp " d1 ", " s1 ", " s1 " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile(
        "s_nop 1\n\t"
        T2_S48("%2", "%3", "%0", "%1", "quad_perm:[1,0,3,2]") "s_nop 0\n\t"
        T2_S48("%0", "%1", "%2", "%3", "quad_perm:[2,3,0,1]") "s_nop 0\n\t"
        T2_S48("%2", "%3", "%0", "%1", "row_half_mirror") "s_nop 0\n\t"
        "v_add_u32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_dpp %1, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(lo), "+v"(hi), "=&v"(a), "=&v"(b));
#undef T2_S48
    return __builtin_fma((double)(int)hi, 16777216.0, (double)lo);
}

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

// ---- carrier tables (T5): entry `lane` of B | W1 | W2 | W3 (lanes 48..63 all hold this member's W3) ----
// table index multiplier: the entry is the phasor of sample m of the block
__device__ __forceinline__ int t2_carr_mult(int lane, int unit, int P, int head) {
    const int sel = lane >> 4, idx = lane & 15;
    return (sel == 3) ? (TRK_UNIT * (unit + idx * P) - head) : (idx << (4 * sel));
}

// (cos, sin)(w m / fs [+ rc for W3]) evaluated in full: fp64 "turns" reduction with w / (2 pi fs) as a double-double
__device__ __forceinline__ void t2_carr_entry(double inv_2pifs_hi, double inv_2pifs_lo, double inv_2pi, double w, double rc,
                                              int mi, bool w3, double& cs, double& sn) {
    const double r_hi = w * inv_2pifs_hi;
    const double r_lo = __builtin_fma(w, inv_2pifs_hi, -r_hi) + w * inv_2pifs_lo;
    const double mult = (double)mi;
    const double pp = r_hi * mult;
    const double ee = __builtin_fma(r_hi, mult, -pp) + r_lo * mult;
    double u = (pp - floor(pp)) + ee;
    const double u3 = u + rc * inv_2pi;        // < 2
    u = w3 ? (u3 - ((u3 >= 1.0) ? 1.0 : 0.0)) : u;
    sgx_sincos_turns_short(u, sn, cs);
}

// Flags in LDS that another wave of the workgroup writes: read and written by LDS instructions, never through a generic
// pointer (a volatile access to a __shared__ object goes through the flat aperture - slower, and one shape of it makes
// this compiler emit an illegal compare against src_shared_base).  The low dword of a generic LDS address is the LDS offset.
__device__ __forceinline__ int lds_peek(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long long)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke(int* p, int v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(unsigned long long)p), "v"(v) : "memory");
}
__device__ __forceinline__ long long lds_peek64(const long long* p) {
    long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long long)p) : "memory");
    return v;
}

#define T2_PIN(x) asm volatile("" : "+v"(x))
#define T2_USE(x) asm volatile("" : : "v"(x))   // the value is needed HERE: its load is not sunk below a later branch

// Everything a role needs that lives in LDS.
struct T2Shared {
    unsigned chip[1032];            // chip[k + 1] = chip of extended-code index k (tracking.py:111), exact-search path
    unsigned cbits[40];             // the same as packed sign bits: bit k + 1 set where chip k is -1
    T2Code code[2];
    T2Carr carr[2];
    double part[2][16][8];          // ARMS = 3: row sums of the map waves by block parity: [wave * 4 + row][word]
    unsigned long long acc[2][2];   // ARMS = 1: {arrival count << 56 | 48-bit fixed-point sum} of I, Q by block parity
    unsigned ticket[2][2];          // ARMS = 3: arrival ticket of the map waves, by block parity
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
typedef double t2_v2d __attribute__((ext_vector_type(2)));
template <int SB> struct T2Raw;
template <> struct T2Raw<1> { uint4 a; };
template <> struct T2Raw<2> { uint4 a, b; };
template <> struct T2Raw<4> { uint4 q[4]; };
template <> struct T2Raw<8> { uint4 q[8]; };

// the (clamped) addresses of a lane's 16 samples, and the load from them: the arm-split map prepares the addresses in the
// shadow of the previous block, so that the chain only holds the load itself
// (byte offsets, not pointers: a pointer kept in a struct loses its address space and the load becomes a flat one)
template <int SB> struct T2Ptr;
template <> struct T2Ptr<1> { long long a; };
template <> struct T2Ptr<2> { long long a, b; };

template <int SB>
__device__ __forceinline__ T2Ptr<SB> t2_ptr(long long first_sample, long long limit) {
    T2Ptr<SB> p;
    const long long a = first_sample * SB;
    p.a = a > limit ? limit : a;
    if constexpr (SB == 2) p.b = a + 16 > limit ? limit : a + 16;
    return p;
}

template <int SB>
__device__ __forceinline__ T2Raw<SB> t2_load_at(const int8_t* __restrict__ rec, const T2Ptr<SB>& p) {
    T2Raw<SB> r;
    r.a = *reinterpret_cast<const uint4*>(rec + p.a);
    if constexpr (SB == 2) r.b = *reinterpret_cast<const uint4*>(rec + p.b);
    return r;
}

template <int SB>
__device__ __forceinline__ T2Raw<SB> t2_load(const int8_t* __restrict__ rec, long long first_sample, long long limit) {
    T2Raw<SB> r;
    if constexpr (SB == 1) {
        r.a = load_group(rec, first_sample, limit);
    } else if constexpr (SB == 2) {
        r.a = load_group(rec, first_sample * 2, limit);
        r.b = load_group(rec, first_sample * 2 + 16, limit);
    } else {
#pragma unroll
        for (int k = 0; k < SB; ++k) r.q[k] = load_group(rec, first_sample * SB + 16 * k, limit);
    }
    return r;
}

// sample b of a lane's 16 as an integer; uns: one-byte samples are unsigned (dataType 'uint8')
template <int SB>
__device__ __forceinline__ int t2_sample(const T2Raw<SB>& raw, int b, bool uns) {
    if constexpr (SB == 1) {
        const unsigned w = (b < 4) ? raw.a.x : (b < 8) ? raw.a.y : (b < 12) ? raw.a.z : raw.a.w;
        const int sh = 8 * (b & 3);
        const int sx = (sh == 24) ? ((int)w >> 24) : (int)(signed char)((w >> sh) & 0xFF);
        return uns ? (sx & 0xFF) : sx;
    } else {
        const uint4& q = (b < 8) ? raw.a : raw.b;
        const int bb = b & 7;
        const unsigned w = (bb < 2) ? q.x : (bb < 4) ? q.y : (bb < 6) ? q.z : q.w;
        return (bb & 1) ? ((int)w >> 16) : (int)(short)(w & 0xFFFF);
    }
}

// 16 samples -> fp64, samples before the block's first one (i0 + b < 0) zeroed; float samples times fscale (a power of two)
template <int SB>
__device__ __forceinline__ void t2_convert(const T2Raw<SB>& raw, int i0, bool uns, double (&xd)[16], double fscale = 1.0) {
    if constexpr (SB == 4) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint4& q = raw.q[b >> 2];
            const unsigned w = ((b & 3) == 0) ? q.x : ((b & 3) == 1) ? q.y : ((b & 3) == 2) ? q.z : q.w;
            xd[b] = (i0 + b >= 0) ? (double)__uint_as_float(w) * fscale : 0.0;
        }
    } else if constexpr (SB == 8) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint4& q = raw.q[b >> 1];
            const double v = (b & 1) ? __hiloint2double((int)q.w, (int)q.z) : __hiloint2double((int)q.y, (int)q.x);
            xd[b] = (i0 + b >= 0) ? v * fscale : 0.0;
        }
    } else {
        (void)fscale;
#pragma unroll
        for (int b = 0; b < 16; ++b) xd[b] = (i0 + b >= 0) ? (double)t2_sample<SB>(raw, b, uns) : 0.0;
    }
}

// 16 samples -> the HIGH dwords of their fp64 values (small integers: the low dword is zero); samples outside the
// block [0, cut) are zeroed
template <int SB>
__device__ __forceinline__ void t2_convert_hi(const T2Raw<SB>& raw, int i0, int cut, bool uns, unsigned (&xh)[16]) {
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const unsigned hi = (unsigned)__double2hiint((double)t2_sample<SB>(raw, b, uns));
        xh[b] = ((unsigned)(i0 + b) < (unsigned)cut) ? hi : 0u;
    }
}

