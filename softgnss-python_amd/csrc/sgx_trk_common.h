// Device helpers shared by the tracking kernels (sgx_trk.hip, sgx_trk2.hip, sgx_trk_tp.hip).  Everything here follows the reference's fp64 operation order where an integer
// rounding follows; both files are built with -ffp-contract=off.
#pragma once
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>

#include "sgx_internal.h"
#include "sgx_trk_math.h"

#define TRK_THREADS 256
#define TRK_UNIT (TRK_THREADS * 16)          // samples per unit (a power of two)
#define TRK_MAX_SPLIT 10                     // 12 granules per member, two per gathering lane

// -DTRK_FINEPROF: fine-grained phase timestamps of (member 0, lane 0); forces waits at every probe, so it
// is a diagnosis build only (tools/ notes in DESIGN.md); the normal build compiles the probes away.
#ifdef TRK_FINEPROF
#define PROBE(k)                                                            \
    do {                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const long long t_ = (long long)__builtin_amdgcn_s_memtime();      \
        fp[k] += t_ - fp_last;                                              \
        fp_last = t_;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#else
#define PROBE(k) do { } while (0)
#endif

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
    int split;            // workgroups cooperating on one channel
    int n_units;          // units that can hold the longest block
    int fast_xcd;         // allow the same-XCD exchange path (SGX_TRK_FASTX=0 disables it)
    int nb_base;          // inv_nb[k] = RN(1 / (nb_base + k)): reciprocals of the possible block lengths
    double inv_nb[8];
    double inv_fs;        // RN(1 / fs)
    double inv_pi;        // RN(1 / pi)
    const unsigned long long* mark;   // streaming record: bytes resident so far (device watermark), or null
    int multi;            // fewer than ~15 samples per chip: a 16-sample group can hold several chip switches
    int uns;              // one-byte samples are unsigned (Settings.dataType 'uint8')
    int kind;             // SGX_DT_* of the record's samples (read by trk_kernel_any; the other kernels are typed)
    double fscale;        // float records on the typed kernel (sgx_trk2.hip): the power of two the samples are scaled by
};

// (struct TrkChan: sgx_internal.h - the device-side preRun of sgx_acq.hip fills it too)

// Per-block parameters: code part written by wave 1, carrier part by wave 0, read by everybody.
struct TrkBlock {
    long long pos;
    int blk;
    int stop;
    double startE, stepE, startP, stepP, startL, stepL;
    double inv_step;          // ~ 1/step, only used to estimate switch samples
    // carrier phasors (cos, sin)(2 pi r m), r = turns per sample:
    double2 B[16];            // m = b                      sample b inside a group
    double2 W1[16];           // m = 16 a                   group a = tid & 15
    double2 W2[16];           // m = 256 b                  group row b = tid >> 4
    double2 W3[16];           // m = 4096 u - head, plus the block's start phase: unit u
};

// Loop state (LDS): code part owned by wave 1, carrier part by wave 0.
struct TrkState {
    double codeFreq, remCode, oldCodeNco, oldCodeErr;
    long long pos;
    double carrFreq, carrBasis, remCarr, w, oldCarrNco, oldCarrErr;
};

// a / b correctly rounded, given y = RN(1/b): reciprocal multiply plus two FMA corrections (Markstein).
// Bit-identical to IEEE division for the divisors used here (pi, fs, block lengths) - checked against exact
// rational arithmetic in tests/test_cabi_and_host.py - at a quarter of the dependent latency of v_div_*.
__device__ __forceinline__ double div_rn(double a, double b, double y) {
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

__device__ __forceinline__ double ramp_at(int i, double step, double start) {
    return (double)i * step + start;   // two roundings, like numpy's y = arange*step; y += start
}

// chip index at sample ilo and first sample whose chip index is larger (exact reference arithmetic).
// (A one-FMA estimate guarded by near-integer tests was tried: it needs a full-precision 1/step and
// measured slower than these two exact probes.)
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
// (rec_len: the record's length in samples when it is not K.rec_len - trk_kernel_any's per-channel sample grid)
__device__ __forceinline__ void prep_code(const TrkConst& K, double codeFreq, double rem, long long pos, TrkState& s,
                                          TrkBlock& b, bool writer, long long rec_len = -1) {
    const double step = div_rn(codeFreq, K.fs, K.inv_fs);                    // T1: codeFreq / fs
    const int blk = sgx_ceil_div(K.code_len - rem, step);                    // == (int)ceil((1023 - rem) / step), always
    const double nb = (double)blk;
    const double span = nb * step;                                           // blksize * codePhaseStep
    // T3: np.linspace(start, stop, blk, endpoint=False): delta = stop - start; stepL = delta / blk
    const int ki = blk - K.nb_base;
    const bool known = (ki >= 0 && ki < 8);                                  // reciprocal of blk precomputed?
    const double ynb = known ? K.inv_nb[ki] : 0.0;
    const double startE = rem - K.spacing;
    const double dE = ((span + rem) - K.spacing) - startE;
    const double startL = rem + K.spacing;
    const double dL = ((span + rem) + K.spacing) - startL;
    const double dP = (span + rem) - rem;
    const double stepE = known ? div_rn(dE, nb, ynb) : dE / nb;
    const double stepL = known ? div_rn(dL, nb, ynb) : dL / nb;
    const double stepP = known ? div_rn(dP, nb, ynb) : dP / nb;
    const double t_last = ramp_at(blk - 1, stepP, rem);
    if (writer) {
        b.pos = pos;
        b.blk = blk;
        b.stop = (blk <= 0 || pos + blk > (rec_len >= 0 ? rec_len : K.rec_len)) ? 1 : 0;
        b.startE = startE;
        b.stepE = stepE;
        b.startL = startL;
        b.stepL = stepL;
        b.startP = rem;
        b.stepP = stepP;
        // 1/step: hardware reciprocal estimate + one Newton step; only used to ESTIMATE switch samples
        const double r0 = __builtin_amdgcn_rcp(step);
        b.inv_step = __builtin_fma(r0, __builtin_fma(-step, r0, 1.0), r0);
        s.remCode = (t_last + step) - 1023.0;                                // T4
        s.pos = pos + blk;
    }
}

// sin and cos of 2 pi u for u in [0, 2): quarter-turn reduction (exact), Taylor polynomials on |theta| <= pi/4.
// ~1 ulp; a short dependent chain matters here because this sits on the per-block critical path.
__device__ __forceinline__ void sincos_turns(double u, double& sn, double& cs) {
    const double q = rint(u * 4.0);
    const double f = __builtin_fma(q, -0.25, u);          // exact, |f| <= 1/8
    const int qi = (int)q & 3;
    const double th = f * 6.283185307179586476925287;
    const double t2 = th * th;
    double ps = -2.8114572543455206e-15;                   // -1/17!
    ps = __builtin_fma(ps, t2, 7.6471637318198164e-13);    //  1/15!
    ps = __builtin_fma(ps, t2, -1.6059043836821613e-10);   // -1/13!
    ps = __builtin_fma(ps, t2, 2.5052108385441720e-08);    //  1/11!
    ps = __builtin_fma(ps, t2, -2.7557319223985893e-06);   // -1/9!
    ps = __builtin_fma(ps, t2, 1.9841269841269841e-04);    //  1/7!
    ps = __builtin_fma(ps, t2, -8.3333333333333332e-03);   // -1/5!
    ps = __builtin_fma(ps, t2, 1.6666666666666666e-01);    //  1/3!  (sign folded below)
    double pc = 4.7794773323873853e-14;                    //  1/16!
    pc = __builtin_fma(pc, t2, -1.1470745597729725e-11);   // -1/14!
    pc = __builtin_fma(pc, t2, 2.0876756987868100e-09);    //  1/12!
    pc = __builtin_fma(pc, t2, -2.7557319223985888e-07);   // -1/10!
    pc = __builtin_fma(pc, t2, 2.4801587301587302e-05);    //  1/8!
    pc = __builtin_fma(pc, t2, -1.3888888888888889e-03);   // -1/6!
    pc = __builtin_fma(pc, t2, 4.1666666666666664e-02);    //  1/4!
    pc = __builtin_fma(pc, t2, -0.5);                      // -1/2!
    const double s0 = __builtin_fma(-(ps * t2), th, th);   // th - th^3 * (1/3! - ...)
    const double c0 = __builtin_fma(pc, t2, 1.0);
    sn = (qi == 0) ? s0 : (qi == 1) ? c0 : (qi == 2) ? -s0 : -c0;
    cs = (qi == 0) ? c0 : (qi == 1) ? -s0 : (qi == 2) ? -c0 : s0;
}

// ---- filter phase, carrier side (wave 0): phasor tables of a block with rate w, start phase remCarr and
// `head` bytes between the 16-byte boundary and the block's first sample -----------------------------------
__device__ __forceinline__ void prep_carr(const TrkConst& K, double w, double remCarr, int head, TrkBlock& b,
                                          int lane) {
    // trigarg = w * (i/fs) + remCarr (T5); in turns: r*i + remCarr/(2 pi), r = w/(2 pi fs) as a double-double
    const double r_hi = w * K.inv_2pifs_hi;
    const double r_lo = __builtin_fma(w, K.inv_2pifs_hi, -r_hi) + w * K.inv_2pifs_lo;
    const int sel = lane >> 4, idx = lane & 15;
    const double mult = (sel == 0) ? (double)idx
                      : (sel == 1) ? (double)(16 * idx)
                      : (sel == 2) ? (double)(256 * idx)
                                   : (double)(TRK_UNIT * idx - head);
    const double p = r_hi * mult;
    const double e = __builtin_fma(r_hi, mult, -p) + r_lo * mult;
    double u = (p - floor(p)) + e;
    if (sel == 3) {
        u += remCarr * K.inv_2pi;        // < 1
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

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(ohi, olo);
}

// sums over lanes 0..31 and over lanes 32..63 of a wave (fixed order, deterministic);
// lanes 0..31 return the first sum, lanes 32..63 the second
__device__ __forceinline__ double half_wave_sum(double v, int lane) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror: every lane of a row of 16 now holds the row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return lane < 32 ? (r0 + r1) : (r2 + r3);
}

// sum over each row of 16 lanes; every lane of a row gets its row's sum (fixed order)
__device__ __forceinline__ double row_sum(double v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return v;
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

// granule store: `fast` = every member of the channel runs on the same XCD (verified at kernel start), so a
// plain store that stays in the shared L2 is visible to the others' L1-bypassing loads; otherwise a
// write-through (agent-scope) store.  Either way ONE aligned 8-byte store per granule.
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v, bool fast) {
    if (fast)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// streaming record: wait until the first `need` bytes are resident (bounded; a stalled loader flags the channel).
// `seen` caches the last watermark read: it only moves in 32 MiB steps, so the (slow, uncached) load is issued
// once per several hundred blocks and not once per block.
#define TRK_ERR_STREAM 0x40000000   // error word: the watermark of a streaming record did not advance in time
#define TRK_ERR_RANGE 0x20000000    // error word: a block is longer than the units the launch provides
#define TRK_ERR_SCALE 0x10000000    // error word: samples too strong for the speculative kernel's 2^30 fixed point (sgx_trk3.hip)

__device__ __forceinline__ void wait_mark(const unsigned long long* mark, long long need, unsigned long long& seen,
                                          int* err, int ch) {
    if ((unsigned long long)need <= seen) return;
    int budget = 1 << 19;   // about a second
    for (;;) {
        seen = __hip_atomic_load(mark, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= (unsigned long long)need) break;
        if (--budget == 0) {
            // give up for good (the host repeats the launch once the whole record is resident): no further waits
            atomicOr(err, TRK_ERR_STREAM);
            (void)ch;
            seen = ~0ull;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// ---- trk_kernel_any (sgx_trk_any.hip): one sample of any little-endian numpy type, at any byte address -------------
template <typename T> struct __attribute__((packed, aligned(1))) AnyAt { T v; };
__host__ __device__ __forceinline__ int sgx_dt_bytes(int kind) {
    switch (kind) {
    case SGX_DT_INT8: case SGX_DT_UINT8: return 1;
    case SGX_DT_INT16: case SGX_DT_UINT16: case SGX_DT_FLOAT16: return 2;
    case SGX_DT_INT32: case SGX_DT_UINT32: case SGX_DT_FLOAT32: return 4;
    case SGX_DT_INT64: case SGX_DT_UINT64: case SGX_DT_FLOAT64: return 8;
    default: return 0;
    }
}
// the value numpy's float64 arithmetic sees (tracking.py:195-196: carrier (float64) * rawSignal promotes every one of
// these types to float64; 64-bit integers round to nearest like numpy's cast)
__device__ __forceinline__ double any_sample(const int8_t* __restrict__ p, int kind) {
    switch (kind) {
    case SGX_DT_INT8: return (double)*p;
    case SGX_DT_UINT8: return (double)*reinterpret_cast<const uint8_t*>(p);
    case SGX_DT_INT16: return (double)reinterpret_cast<const AnyAt<short>*>(p)->v;
    case SGX_DT_UINT16: return (double)reinterpret_cast<const AnyAt<unsigned short>*>(p)->v;
    case SGX_DT_INT32: return (double)reinterpret_cast<const AnyAt<int>*>(p)->v;
    case SGX_DT_UINT32: return (double)reinterpret_cast<const AnyAt<unsigned>*>(p)->v;
    case SGX_DT_INT64: return (double)reinterpret_cast<const AnyAt<long long>*>(p)->v;
    case SGX_DT_UINT64: return (double)reinterpret_cast<const AnyAt<unsigned long long>*>(p)->v;
    case SGX_DT_FLOAT16: return (double)__half2float(reinterpret_cast<const AnyAt<__half>*>(p)->v);
    case SGX_DT_FLOAT32: return (double)reinterpret_cast<const AnyAt<float>*>(p)->v;
    default: return reinterpret_cast<const AnyAt<double>*>(p)->v;
    }
}

// the 16 samples of a group (sample i0 + b of the block at p + b * sizeof(T)); samples outside [0, blk) are zero and not read
template <typename T>
__device__ __forceinline__ void any_group_t(const int8_t* __restrict__ p, int i0, int blk, double (&xd)[16]) {
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        T v = T(0);
        if ((unsigned)(i0 + b) < (unsigned)blk) v = reinterpret_cast<const AnyAt<T>*>(p + (long long)b * (long long)sizeof(T))->v;
        xd[b] = (double)v;
    }
}
__device__ __forceinline__ void any_group(const int8_t* __restrict__ p, int kind, int i0, int blk, double (&xd)[16]) {
    switch (kind) {                      // (wave-uniform: one branch per group, not per sample)
    case SGX_DT_INT8: any_group_t<signed char>(p, i0, blk, xd); break;
    case SGX_DT_UINT8: any_group_t<unsigned char>(p, i0, blk, xd); break;
    case SGX_DT_INT16: any_group_t<short>(p, i0, blk, xd); break;
    case SGX_DT_UINT16: any_group_t<unsigned short>(p, i0, blk, xd); break;
    case SGX_DT_INT32: any_group_t<int>(p, i0, blk, xd); break;
    case SGX_DT_UINT32: any_group_t<unsigned>(p, i0, blk, xd); break;
    case SGX_DT_INT64: any_group_t<long long>(p, i0, blk, xd); break;
    case SGX_DT_UINT64: any_group_t<unsigned long long>(p, i0, blk, xd); break;
    case SGX_DT_FLOAT32: any_group_t<float>(p, i0, blk, xd); break;
    case SGX_DT_FLOAT16: {
#pragma unroll
        for (int b = 0; b < 16; ++b)
            xd[b] = ((unsigned)(i0 + b) < (unsigned)blk) ? (double)__half2float(reinterpret_cast<const AnyAt<__half>*>(p + 2 * b)->v) : 0.0;
        break;
    }
    default: any_group_t<double>(p, i0, blk, xd); break;
    }
}

__device__ __forceinline__ uint4 load_group(const int8_t* __restrict__ rec, long long addr, long long limit) {
    if (addr > limit) addr = limit;   // never read past the allocation (data of a stopped block is unused)
    return *reinterpret_cast<const uint4*>(rec + addr);
}

// ---- chip-lane maps (sgx_trk_tp.hip: one lane per prompt chip; sgx_trk_chip.hip: one lane per half chip) ----
#define TP_RUN 20   // samples per run handled by the unrolled path

// first sample i with T(i) = fl(fl(i*step)+start) > thr (exact reference arithmetic; see ramp_setup)
__device__ __forceinline__ int first_above(double start, double step, double inv_step, double thr) {
    const int cand = (int)ceil((thr - start) * inv_step);
    const bool at0 = ramp_at(cand, step, start) > thr;
    const bool atm = ramp_at(cand - 1, step, start) > thr;
    return at0 ? (atm ? cand - 1 : cand) : cand + 1;
}

// carrier tables for block-relative sample indices: phasor(i) = W3[i>>12] * W2[(i>>8)&15] * W1[(i>>4)&15] * B[i&15]
struct TpCarr {
    double2 B[32];    // (cos, sin)(2 pi r k), k = 0..31
    double2 W1[16];   // k = 16 a
    double2 W2[16];   // k = 256 b
    double2 W3[16];   // k = 4096 u, plus the block's start phase
};

__device__ __forceinline__ void tp_tables(const TrkConst& K, double w, double remCarr, TpCarr& t, int lane, int round) {
    const double r_hi = w * K.inv_2pifs_hi;
    const double r_lo = __builtin_fma(w, K.inv_2pifs_hi, -r_hi) + w * K.inv_2pifs_lo;
    double mult;
    if (round == 0)
        mult = (lane < 32) ? (double)lane : (lane < 48) ? (double)(16 * (lane - 32)) : (double)(256 * (lane - 48));
    else
        mult = (double)(4096 * (lane & 15));
    const double p = r_hi * mult;
    const double e = __builtin_fma(r_hi, mult, -p) + r_lo * mult;
    double u = (p - floor(p)) + e;
    if (round == 1) {
        u += remCarr * K.inv_2pi;
        u -= (u >= 1.0) ? 1.0 : 0.0;
    }
    double sn, cs;
    sincos_turns(u, sn, cs);
    const double2 v = make_double2(cs, sn);
    if (round == 0) {
        if (lane < 32) t.B[lane] = v;
        else if (lane < 48) t.W1[lane - 32] = v;
        else t.W2[lane - 48] = v;
    } else if (lane < 16) {
        t.W3[lane] = v;
    }
}

__device__ __forceinline__ double2 cmul2(double2 a, double2 b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}

struct __attribute__((packed, aligned(4))) U4a { unsigned x, y, z, w; };   // dword-aligned 16-byte load
struct __attribute__((packed, aligned(4))) U2a { unsigned x, y; };

// 20 bytes starting at record byte `addr` (any alignment), bytes >= len zeroed: five dwords
__device__ __forceinline__ void load_run(const int8_t* __restrict__ rec, long long addr, long long limit, int len,
                                         unsigned (&w)[5]) {
    long long a4 = addr & ~3ll;
    if (a4 > limit) a4 = limit;
    const unsigned sh = (unsigned)(addr & 3);
    const U4a q = *reinterpret_cast<const U4a*>(rec + a4);
    const U2a q2 = *reinterpret_cast<const U2a*>(rec + a4 + 16);
    w[0] = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
    w[1] = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
    w[2] = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
    w[3] = __builtin_amdgcn_alignbyte(q2.x, q.w, sh);
    w[4] = __builtin_amdgcn_alignbyte(q2.y, q2.x, sh);
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        int keep = len - 4 * d;                      // bytes of this dword inside the run
        keep = keep < 0 ? 0 : (keep > 4 ? 4 : keep);
        w[d] &= (keep >= 4) ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u);
    }
}

