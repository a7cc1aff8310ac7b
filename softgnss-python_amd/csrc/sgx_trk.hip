// TrackingResult.track on gfx950 (reference tracking.py:13-295): the host side.
//
// A channel is a chain of `ms` dependent 1-ms steps: every block's length, code ramps and NCO rates depend on the
// previous block's six correlator sums.  Channels are independent.  Every kernel is persistent: one launch walks all
// code periods of all channels.  A block (~38 192 samples) is cut into UNITS of 256 groups x 16 samples (4 KiB of IF).
//
// Three kernels, chosen here:
//   trk2_kernel     (sgx_trk2.hip) - the latency-mode kernel, every cooperative case and its own fallback: P members per
//                   channel, member m owns units m, m + P, ...  P = units (one workgroup per unit - or, when three times
//                   as many CUs are free, one per unit and correlator arm) while the CUs last, fewer members with
//                   several units each for more channels, P = 1 (no co-residency needed) when a cooperative launch
//                   timed out or the CUs are taken.  int8 and int16 records, resident or still streaming in.
//   trk_kernel_tp   (sgx_trk_tp.hip) - throughput mode: more than 128 int8 channels, one workgroup per channel.
//   trk_kernel_multi (sgx_trk_multi.hip) - sampling rates below ~15.4 samples per chip, where a 16-sample group can hold
//                   several chip switches of one ramp (per-sample replica lookup; the round-1 cooperative body).
// fp64 everywhere: 1e-7 errors in the sums move the code NCO enough to flip a chip-boundary sample somewhere in a 37 s
// run, which is a 1e-3 relative blip.
#include <chrono>
#include "sgx_trk_common.h"

// sgx_trk_tp.hip: throughput-mode kernel (one lane per prompt chip, two workgroups per CU) for split == 1, > 128 channels
void sgx_trk_tp_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                       double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err);

// sgx_trk2.hip: the latency-mode kernel (one unit per member - or, arms = 1, one unit and one correlator arm per
// member -, tagged-granule exchange, dedicated filter waves)
void sgx_trk2_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                     double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err,
                     int sample_bytes, int arms, int lds_pad);
#define T2_MAXP 16
#ifndef T2_XLINE
#define T2_XLINE 16
#endif
#define T2_XCH_STRIDE (((12 * T2_XLINE + 8 + 48) + 255) / 256 * 256)   // (as in sgx_trk2.hip)
#define T2_PROF_STRIDE 192

// sgx_trk3.hip: the speculative latency-mode kernel (round 4): one workgroup per unit of 128 groups serves all three
// correlator arms, the map runs one block ahead of the loop filter and only corrections are on the per-block chain
void sgx_trk3_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                     double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch, int* err,
                     int lds_pad);
#define T3_LANES 128
#define T3_MAXP 32
#ifndef T3_XCH_STRIDE
#define T3_XCH_STRIDE 512   // (as in sgx_trk3.hip)
#endif

// sgx_trk_multi.hip: the cooperative kernel with a per-sample replica lookup, for low sampling rates
void sgx_trk_multi_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                          double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                          int* err);

// sgx_trk_any.hip: the same body with every sample fetched where it lies, for any sample type (K.kind)
void sgx_trk_any_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
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

// What a tracking launch leaves for the host (round 6): the error words and every channel's ms_done, copied to the pinned
// result page by ONE small kernel behind the tracking kernel, then the word the host spins on - instead of two copies to
// pageable memory with a stream synchronisation each (~80 us behind every launch).
struct TrkLook {
    unsigned long long seq;
    int err[2];
    int done[SGX_TRK_LOOK_CH];
};
static_assert(sizeof(TrkLook) <= SGX_LOOK_BYTES - SGX_TRK_LOOK_OFFSET, "the tracking look fits its part of the page");
__global__ __launch_bounds__(SGX_TRK_LOOK_CH) void trk_finish_kernel(const int* __restrict__ d_err, const int* __restrict__ d_done,
                                                                     int n_ch, TrkLook* __restrict__ look, unsigned long long seq) {
    const int t = threadIdx.x;
    if (t < 2) look->err[t] = d_err[t];
    if (t < n_ch) look->done[t] = d_done[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&look->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// THE SPECULATIVE KERNEL'S SCALE GUARD for resident int8 records (sgx_trk3.hip: a unit's total must stay below 2^17, and no
// arm's total can exceed the sum of the unit's magnitudes).  One pass over the record, once per record (cached in the
// handle): the largest sum of |x| over 17 consecutive 128-byte blocks - any 2 048-byte window of the kernel lies inside
// such a run - by workgroups of 256 blocks with a halo of 16.  1.4 GB in ~0.4 ms; a streaming record that is not resident
// yet is guarded inside the kernel instead (its record wave adds up the magnitudes of every block's window).
__global__ __launch_bounds__(256) void if_mag_kernel(const int8_t* __restrict__ x, long long n_bytes, int* __restrict__ out_max) {
    __shared__ int s_m[256 + 16];
    const long long blk0 = (long long)blockIdx.x * 256;
    auto block_mag = [&](long long j) -> int {
        const long long a = j * 128;
        if (a >= n_bytes) return 0;                      // (the allocation is padded with zero bytes: SGX_IF_PAD)
        int m = 0;
        const uint4* p = reinterpret_cast<const uint4*>(x + a);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 w = p[k];
            const unsigned b = 0x80808080u;
            m = (int)__builtin_amdgcn_sad_u8(w.x ^ b, b, (unsigned)m);
            m = (int)__builtin_amdgcn_sad_u8(w.y ^ b, b, (unsigned)m);
            m = (int)__builtin_amdgcn_sad_u8(w.z ^ b, b, (unsigned)m);
            m = (int)__builtin_amdgcn_sad_u8(w.w ^ b, b, (unsigned)m);
        }
        return m;
    };
    s_m[threadIdx.x] = block_mag(blk0 + threadIdx.x);
    if (threadIdx.x < 16) s_m[256 + threadIdx.x] = block_mag(blk0 + 256 + threadIdx.x);
    __syncthreads();
    int w = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) w += s_m[threadIdx.x + k];
    for (int o = 32; o > 0; o >>= 1) {
        const int v = __shfl_down(w, o);
        w = v > w ? v : w;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_max, w);
}

// -> the bound (cached in the handle), or -1 when the record is not fully resident yet / on an error
static long long if_mag_bound(sgx_ctx* c, const sgx_if* r) {
    long long known = r->mag_max.load();
    if (known >= 0) return known;
    if (r->loader && !r->load_done.load()) return -1;
    int* d_max = (int*)((char*)c->d_small + 730000);
    if (hipMemsetAsync(d_max, 0, sizeof(int), c->stream) != hipSuccess) return -1;
    const long long n_blocks128 = ((long long)r->n + 127) / 128;
    const unsigned grid = (unsigned)((n_blocks128 + 255) / 256);
    if_mag_kernel<<<grid ? grid : 1u, 256, 0, c->stream>>>(r->d, (long long)r->n + SGX_IF_PAD - 128, d_max);
    int h = 0;
    if (hipMemcpyAsync(&h, d_max, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    const_cast<sgx_if*>(r)->mag_max.store((long long)h);
    return (long long)h;
}

// ---- WHICH KERNEL, HOW MANY MEMBERS: the one rule (include/sgx.h: sgx_track_plan; tests/test_cabi_and_host.py holds the table)
//
//   sample type                      channels   samples / chip      spacing   -> kernel                        members per channel
//   any but int8/uint8/int16/float*  any        any                 any          6 trk_kernel_any              min(units, CUs / ch8, 10)
//   float32 / float64 out of range   any        any                 any          6 trk_kernel_any              (as above)
//   uint8 / int16 / float            any        < ~15.4 (multi)     any          6 trk_kernel_any              (as above)
//   int8                             any        < ~15.4 (multi)     any          4 trk_kernel_multi            min(units, CUs / ch8, 10)
//   int8 / uint8 / int16             > 128 and CUs / ch8 < 2        any          3 trk_kernel_tp               1
//   int8 / uint8                     ch8 x 2 units <= CUs, 18 samples < 1/2 chip, spacing 1/2
//                                                                                5 trk3_kernel                 2 x units (128-group units)
//   everything else                                                              2 trk2_kernel                 3 x units (one per unit and arm)
//                                                                                                              while 3 ch8 units <= CUs and not float,
//                                                                                                              else min(units, CUs / ch8)
//   (ch8 = channels rounded up to 8; units = ceil((samplesPerCode + 94) / 16 / 256); a cooperative launch that cannot be
//   resident, or whose member timed out, is repeated with ONE member per channel by sgx_track_kind)
struct TrkPlan {
    int kernel, members;
    int multi, use_any, use_tp, use_v2, use_v3, arm_split, split, n_units, n_units3;
};

// (split_env: SGX_TRK_SPLIT or 0; arms_env: 0 unset, 3 SGX_TRK_ARMS=3, 1 any other value; v3_off: SGX_TRK_V3=0 - diagnostics)
static TrkPlan trk_plan(const sgx_settings& S, int kind, int n_ch, long long n_code, int cus_total, bool floaty,
                        int split_env = 0, int arms_env = 0, bool v3_off = false) {
    TrkPlan P;
    const int sample_bytes = sgx_dt_bytes(kind);
    P.multi = (15.0 * 1.001 * S.codeFreqBasis / S.samplingFreq >= 1.0) ? 1 : 0;
    if (P.multi) floaty = false;
    const bool typed = kind == SGX_DT_INT8 || kind == SGX_DT_UINT8 || kind == SGX_DT_INT16 || floaty;
    P.use_any = (!typed || (P.multi && kind != SGX_DT_INT8)) ? 1 : 0;
    const int ch8 = ((n_ch + 7) / 8) * 8;
    P.n_units = (int)((n_code + 64 + 15 + 15) / 16 + TRK_THREADS - 1) / TRK_THREADS;
    int split = cus_total / ch8;
    if (split > P.n_units) split = P.n_units;
    if ((P.multi || P.use_any) && split > TRK_MAX_SPLIT) split = TRK_MAX_SPLIT;
    if (split < 1) split = 1;
    if (split_env >= 1 && split_env <= split) split = split_env;
    P.split = split;
    P.use_tp = (!P.use_any && !floaty && !P.multi && split == 1 && n_ch > 128) ? 1 : 0;
    P.use_v2 = (!P.use_any && !P.multi && !P.use_tp) ? 1 : 0;
    P.arm_split = (P.use_v2 && !floaty && split == P.n_units && P.n_units >= 2 && 3 * ch8 * P.n_units <= cus_total &&
                   split_env == 0 && arms_env != 3) ? 1 : 0;
    // The speculative kernel (sgx_trk3.hip) serves all three arms from one lane, which rests on a 16-sample group (and one
    // sample on either side of it) meeting at most ONE chip boundary of ANY arm: the arms' boundaries lie at code phases
    // 0, d and 1 - d (mod 1 chip; d = dllCorrelatorSpacing), so the smallest gap between two DIFFERENT ones must exceed 18
    // samples of code phase (1 % margin for the code NCO) - and its fused half-chip ramp puts the early / late boundaries
    // on the ODD half chips: spacing 1/2 exactly.  int8 / uint8 records, one workgroup per unit of 128 groups, while
    // 8-padded channels x units fit the CUs.
    P.n_units3 = 2 * P.n_units;
    {
        const double d = S.dllCorrelatorSpacing, e = 1.0 - d;
        double pts[3] = {0.0, d < e ? d : e, d < e ? e : d};
        double gap = 2.0;
        for (int i = 0; i < 3; ++i) {
            const double g = (i < 2 ? pts[i + 1] : pts[0] + 1.0) - pts[i];
            if (g > 1e-9 && g < gap) gap = g;
        }
        const double stepn = S.codeFreqBasis / S.samplingFreq;
        P.use_v3 = (P.use_v2 && sample_bytes == 1 && P.n_units3 >= 2 && P.n_units3 <= T3_MAXP && ch8 * P.n_units3 <= cus_total &&
                    fabs(d - 0.5) < 1e-12 && 18.0 * stepn * 1.01 <= gap && split_env == 0 && arms_env == 0 && !v3_off) ? 1 : 0;
    }
    if (P.use_any) { P.kernel = 6; P.members = split; }
    else if (P.use_tp) { P.kernel = 3; P.members = 1; }
    else if (P.use_v3) { P.kernel = 5; P.members = P.n_units3; }
    else if (P.use_v2) { P.kernel = 2; P.members = (P.arm_split && split > 1) ? 3 * split : split; }
    else { P.kernel = 4; P.members = split; }
    return P;
}

extern "C" int sgx_track_plan(const sgx_settings* s, int32_t data_type, int32_t n_ch, int32_t n_cus, int32_t float_in_range,
                              int32_t* kernel, int32_t* members) {
    SGX_CHECK_ARG(s && kernel && members && n_ch >= 1 && n_cus >= 1);
    if (sgx_dt_bytes(data_type) == 0) {
        sgx_set_error("sgx_track_plan: data_type %d is not one of SGX_DT_* (include/sgx.h)", (int)data_type);
        return SGX_E_ARG;
    }
    int64_t n_code = 0;
    const int rc = sgx_samples_per_code(s, &n_code);
    if (rc != SGX_OK) return rc;
    const bool fl = (data_type == SGX_DT_FLOAT32 || data_type == SGX_DT_FLOAT64) && float_in_range != 0;
    const TrkPlan P = trk_plan(*s, data_type, n_ch, (long long)n_code, n_cus, fl);
    *kernel = P.kernel;
    *members = P.members;
    return SGX_OK;
}

// sample_bytes: 1 (int8 record) or 2 (little-endian int16 record; the record handle holds the file's BYTES).  The
// reference seeks skipNumberOfBytes + codePhase BYTES whatever the sample type and reports fid.tell(), also bytes
// (tracking.py:107, 255); so a two-byte channel may start on an odd byte - its samples then straddle the file's - and
// the kernel follows it there (per-channel byte shift of the record pointer, unaligned 16-byte loads).
// kind: SGX_DT_*.  int8 / uint8 / int16 run the typed kernels (sgx_trk2 / sgx_trk3 / sgx_trk_tp); every other type - and
// int16 / uint8 at sampling rates below 16 x the chip rate - the per-sample kernel of sgx_trk_any.hip.
// skip_bytes: Settings.skipNumberOfBytes, or what stands in for it (sgx_trk_f32.hip tracks a narrowed copy of a window).
// fscale > 0 (float32 / float64 only): every sample the channels can reach is finite and at most 128 / fscale in magnitude
// (sgx_trk_f32.hip has scanned the window; fscale is a power of two) - the record then runs the latency-mode kernel, which
// scales the samples by it on conversion; the correlator series are scaled back here.  0: the per-sample kernel.
// chained (round 6, sgx_track_chained): `ch` is null - the channel table is made ON THE DEVICE by the preRun kernel queued
// in front of the first launch (sgx_prerun_enqueue, sgx_acq.hip) from the acquisition that is pending on this context.
static int track_kind_impl(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch,
                           int32_t ms, double* out, int32_t* ms_done, int kind, long long skip_bytes, double fscale,
                           bool chained) {
    SGX_CHECK_ARG(c && r && (ch || chained) && out && ms_done);
    const int sample_bytes = sgx_dt_bytes(kind);
    SGX_CHECK_ARG(sample_bytes >= 1);
    const bool sample_uns = kind == SGX_DT_UINT8;
    SGX_CHECK_ARG(n_ch >= 1 && n_ch <= 65535 && ms >= 1);
    if (!(c->s.dllCorrelatorSpacing > 0.0 && c->s.dllCorrelatorSpacing < 1.0)) {
        // beyond one chip the reference's replica index ceil(t) leaves its 1025-entry code table (or wraps)
        sgx_set_error("dllCorrelatorSpacing %g outside (0, 1) chips", c->s.dllCorrelatorSpacing);
        return SGX_E_ARG;
    }
    SGX_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const sgx_settings& S = c->s;
    // (diagnosis) SGX_STEP_TRACE=1: host-side time stamps of this call on stderr, microseconds since its entry
    const char* tre = getenv("SGX_STEP_TRACE");
    const bool trace = tre && tre[0] == '1';
    const auto tr0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace) fprintf(stderr, "[sgx step trace] %-28s %8.1f us\n", what,
                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count());
    };

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
    K.rec_len = (long long)r->n;                         // bytes; two-byte samples: the kernel divides (per-channel shift)
    K.rec_alloc = (long long)r->n + SGX_IF_PAD - (sample_bytes - 1);   // bytes, less the largest per-channel shift
    K.mark = nullptr;
    // (a float record the typed kernel can take: in range, no channel starting inside a sample, not switched off)
    bool floaty = (kind == SGX_DT_FLOAT32 || kind == SGX_DT_FLOAT64) && fscale > 0.0;
    for (int i = 0; i < n_ch && floaty && !chained; ++i) {
        // (a channel that starts inside a sample reads other values than the ones that were scanned)
        const long long p0 = skip_bytes + (long long)ch[i].codePhase - rec_file_offset;
        if (ch[i].prn != 0 && p0 >= 0 && p0 % sample_bytes != 0) floaty = false;
    }
    {
        const char* fe = getenv("SGX_TRK_FLOAT_TYPED");   // '0': float records always on the per-sample kernel
        if (fe && fe[0] == '0') floaty = false;
    }
    int cus_total = 0;
    SGX_HIP(hipDeviceGetAttribute(&cus_total, hipDeviceAttributeMultiprocessorCount, c->device));
    // THE RULE (trk_plan above), with the diagnostic overrides of this process's environment
    const char* se = getenv("SGX_TRK_SPLIT");
    const char* ae = getenv("SGX_TRK_ARMS");
    const char* v3e = getenv("SGX_TRK_V3");
    const TrkPlan P0 = trk_plan(S, kind, n_ch, (long long)c->n_code, cus_total, floaty, (se && atoi(se) >= 1) ? atoi(se) : 0,
                                ae ? (ae[0] == '3' ? 3 : 1) : 0, v3e && v3e[0] == '0');
    K.multi = P0.multi;
    if (K.multi) floaty = false;
    K.uns = sample_uns ? 1 : 0;
    K.kind = kind;
    K.fscale = floaty ? fscale : 1.0;
    const bool use_any = P0.use_any != 0;
    K.file_off = rec_file_offset;
    K.ms = ms;
    K.n_ch = n_ch;
    const int ch8 = ((n_ch + 7) / 8) * 8;
    {
        // units needed by the longest possible block, worst alignment.  A block is samplesPerCode +- 1 samples long
        // while the code NCO stays near its basis; the allowance of 64 samples corresponds to a code-rate error of
        // 0.17 % (1.7 kHz at 1.023 MHz), three orders of magnitude beyond what the DLL's filter can command.
        K.n_units = P0.n_units;
        if (K.n_units > 16) {
            sgx_set_error("samplesPerCode %lld needs %d units, the tracking kernel holds 16", (long long)c->n_code,
                          K.n_units);
            return SGX_E_ARG;
        }
        // members (cooperating workgroups) per channel: one workgroup per CU, all of a cooperative launch must be
        // resident at once (they wait for each other), so members * channels <= CU count
        K.split = P0.split;
        const char* fe = getenv("SGX_TRK_FASTX");
        K.fast_xcd = (fe && fe[0] == '0') ? 0 : 1;
        K.nb_base = (int)c->n_code - 3;
        for (int k = 0; k < 8; ++k) K.inv_nb[k] = 1.0 / (double)(K.nb_base + k);
        K.inv_fs = 1.0 / S.samplingFreq;
        K.inv_pi = 1.0 / M_PI;
    }

    std::vector<TrkChan> hc((size_t)n_ch);
    for (int i = 0; i < n_ch && !chained; ++i) {
        hc[(size_t)i].acquiredFreq = ch[i].acquiredFreq;
        hc[(size_t)i].prn = ch[i].prn;
        hc[(size_t)i].pad = 0;
        SGX_CHECK_ARG(ch[i].prn >= 0 && ch[i].prn <= 32);
        const long long p0 = skip_bytes + (long long)ch[i].codePhase - rec_file_offset;
        if (ch[i].prn != 0 && p0 < 0) {
            sgx_set_error("channel %d starts at file byte %lld, before the record (offset %lld)", i,
                          skip_bytes + (long long)ch[i].codePhase, (long long)rec_file_offset);
            return SGX_E_RANGE;
        }
        // two-byte samples: the channel's own sample grid starts at byte (p0 & 1) of the record
        hc[(size_t)i].pos0 = p0 / sample_bytes;
        hc[(size_t)i].pad = (int)(p0 % sample_bytes);
    }
    const size_t elems = (size_t)n_ch * SGX_NUM_SERIES * (size_t)ms;
    // If the caller's result buffer is pinned host memory (sgx_host_alloc; the Python binding's is), the kernel's
    // record stores - 104 bytes per channel per millisecond, issued by an otherwise idle wave - go straight to it
    // over PCIe: no device staging buffer, no D2H copy and no prefill pass after the kernel.
    double* d_out = nullptr;
    bool direct = false;
    {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, out) == hipSuccess && pa.type == hipMemoryTypeHost && pa.devicePointer) {
            d_out = (double*)pa.devicePointer;
            direct = true;
        } else {
            (void)hipGetLastError();   // a pageable pointer is not an error
        }
    }
    if (!direct) {
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
        d_out = c->d_trk_out;
    }
    // device-side call state lives in one cached allocation: [channels | done | exchange | err | profile]
    const size_t sz_ch = ((sizeof(TrkChan) * (size_t)n_ch + 255) / 256) * 256;
    const size_t sz_done = ((sizeof(int) * (size_t)n_ch + 255) / 256) * 256;
    size_t xch_words = (2 * TRK_MAX_SPLIT * 12 + 16) > T2_XCH_STRIDE ? (2 * TRK_MAX_SPLIT * 12 + 16) : T2_XCH_STRIDE;
    if (xch_words < T3_XCH_STRIDE) xch_words = T3_XCH_STRIDE;
    const size_t xch_bytes = sizeof(unsigned long long) * (size_t)n_ch * xch_words;
    const size_t sz_xch = ((xch_bytes + 255) / 256) * 256;
    const size_t sz_prof = sizeof(long long) * T2_PROF_STRIDE * (size_t)n_ch;
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
    const char* pe = getenv("SGX_TRK_PROFILE");
    const bool want_prof = pe && pe[0] == '1';
    long long* d_prof = want_prof ? (long long*)(aux + sz_ch + sz_done + sz_xch + 256) : nullptr;
    // Which kernel: the low-rate variant when a group can hold several switches of a ramp; throughput mode for more
    // than 128 int8 channels (one workgroup per channel anyway); the latency-mode kernel otherwise - with one workgroup
    // per (unit, correlator arm) when three times the CUs of one-per-unit are free (SGX_TRK_ARMS=3 keeps one per unit).
    const bool use_tp = P0.use_tp != 0;
    const bool use_v2 = P0.use_v2 != 0;
    const bool arm_split = P0.arm_split != 0;
    const bool use_v3 = P0.use_v3 != 0;
    const int n_units2 = K.n_units;
    const int n_units3 = P0.n_units3;   // (units of half the size: the same room for a code NCO that left its basis)
    const char* le = getenv("SGX_TRK_LDSPAD");   // dynamic LDS per workgroup (bytes); default: one workgroup per CU
    const int lds_pad_coop = le ? atoi(le) : 90112;
    const int split0 = K.split;
    // CUs claimed by a cooperative launch; given back on EVERY way out of this function
    struct CuGuard {
        int device, n;
        ~CuGuard() { drop(); }
        void drop() {
            if (n) sgx_cu_release(device, n);
            n = 0;
        }
    } reserved{c->device, 0};
    bool used_v2 = false;
    // what the launches so far have established: the record's streaming has been tried (and stalled), a member of a
    // cooperative layout timed out (the next launch runs with one workgroup per channel)
    bool stream_tried = false, fallback_one = false, v3_off = false;
    // the launch's error words and ms_done through the pinned page (one small kernel, the host spins) when the series go
    // straight to the caller's pinned buffer and the channels fit the page
    const bool fast_look = direct && n_ch <= SGX_TRK_LOOK_CH && !want_prof;
    const TrkLook* h_look = (const TrkLook*)((const char*)c->h_look + SGX_TRK_LOOK_OFFSET);
    if (use_v3 && kind == SGX_DT_INT8 && !(r->loader && !r->load_done.load())) {
        // (resident int8 record: the scale guard is a bound computed once per record; sgx_trk3.hip says why 2^17)
        const long long mag = if_mag_bound(c, r);
        if (mag >= 131072) {
            fprintf(stderr, "[sgx] tracking: samples too strong for the speculative kernel's fixed point (2 048 samples add up "
                            "to 131 072 or more in magnitude); the round-3 kernel tracks this record\n");
            v3_off = true;
        }
    }
    int used_members = 0;
    hipError_t e = hipSuccess;
    int h_err = 0;
    // The cooperating workgroups of a channel wait for each other, so all of them must be resident at once.  If
    // something else occupies the CUs a member times out (bounded spins) and flags the channel: the launch is
    // then repeated once with one workgroup per channel, which needs no co-residency.  A streaming record whose
    // watermark stalls is repeated on the resident record first, with the same decomposition; a record too strong for the
    // speculative kernel's fixed point is repeated with the round-3 kernel: at most four launches, each repeat said on stderr.
    if (chained && r->loader && !r->load_done.load()) return SGX_E_DEFER;   // (a record that is still streaming in)
    for (int launches = 0; launches < 4; ++launches) {
        if (!chained) {
            SGX_HIP(hipMemcpyAsync(d_ch, hc.data(), sizeof(TrkChan) * (size_t)n_ch, hipMemcpyHostToDevice, st));
        } else if (launches == 0) {
            // preRun on the device: the table lands in d_ch (a repeated launch finds it there)
            const int rp = sgx_prerun_enqueue(c, d_ch, n_ch, skip_bytes, rec_file_offset, sample_bytes);
            if (rp != SGX_OK) return rp;
        }
        stamp("channel table queued");
        SGX_HIP(hipMemsetAsync(aux + sz_ch, 0, sz_done + sz_xch + 256, st));   // done, every polled word, err
        stamp("memset queued");
        if (!direct) trk_fill_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, st>>>(d_out, ms, (long long)elems);
        // a record that is still streaming in is followed by the latency-mode kernel (its record wave watches the
        // device watermark); the other kernels, and a launch repeated on the resident record, first wait for all of it
        const char* se2 = getenv("SGX_TRK_STREAM");
        const bool v2 = use_v2;
        const bool want_stream = r->loader && !r->load_done.load() && launches == 0 && !(se2 && se2[0] == '0') && v2;
        K.split = fallback_one ? 1 : split0;                     // a member timed out: no co-residency needed with one
        K.n_units = n_units2;
        bool v3 = use_v3 && !fallback_one && !v3_off;
        if (v3) {
            reserved.n = sgx_cu_reserve(c->device, cus_total, ch8 * n_units3);
            if (reserved.n == 0) v3 = false;                     // (the CUs are taken: the layouts below need fewer)
            else {
                K.split = n_units3;
                K.n_units = n_units3;
            }
        }
        int arms_now = (v2 && arm_split && K.split > 1) ? 1 : 3;
        // Cooperating workgroups wait for each other, so all of a launch must be resident at once: one workgroup per CU
        // out of a per-device budget shared by every context of this process (a launch that does not fit the CUs left
        // by the others runs with one workgroup per channel, which needs no co-residency).
        if (K.split > 1 && !v3) {
            reserved.n = sgx_cu_reserve(c->device, cus_total, ch8 * K.split * (arms_now == 1 ? 3 : 1));
            if (reserved.n == 0 && arms_now == 1) {
                arms_now = 3;                                    // the CUs left may still hold one workgroup per unit
                reserved.n = sgx_cu_reserve(c->device, cus_total, ch8 * K.split);
            }
            if (reserved.n == 0) K.split = 1;
        }
        const int members_now = v3 ? K.split : K.split * ((v2 && K.split > 1 && arms_now == 1) ? 3 : 1);
        const int n_blocks = ch8 * members_now;
        const bool streaming = want_stream;
        if (!streaming) {
            const int rq = sgx_if_require(r, r->n);
            if (rq != SGX_OK) return rq;
        }
        K.mark = streaming ? r->d_mark : nullptr;
        if (want_prof) SGX_HIP(hipMemsetAsync(d_prof, 0, sz_prof, st));
        hipEventRecord(c->ev[3], st);
        if (v3) {
            const char* wh = getenv("SGX_TRK_TEST_WITHHOLD");   // test hook: launch without each channel's last member
            const int nb3 = (wh && wh[0] == '1') ? n_blocks - 8 : n_blocks;
            sgx_trk3_launch(nb3, st, r->d, c->d_codes, d_ch, d_out, d_done, K, d_prof, d_xch, d_err, lds_pad_coop);
            used_v2 = true;
            used_members = members_now;
            c->timing.track_kernel = 5.f;
        } else if (v2) {
            const char* wh = getenv("SGX_TRK_TEST_WITHHOLD");   // test hook: launch without each channel's last member
            const int nb2 = (wh && wh[0] == '1' && K.split > 1) ? n_blocks - 8 : n_blocks;
            // (one workgroup per CU only matters while members wait for each other)
            sgx_trk2_launch(nb2, st, r->d, c->d_codes, d_ch, d_out, d_done, K, d_prof, d_xch, d_err, sample_bytes, arms_now,
                            K.split > 1 ? lds_pad_coop : 0);
            used_v2 = true;
            used_members = members_now;
            c->timing.track_kernel = 2.f;
        }
        else if (use_any) {  // (any sample type, sample by sample)
            const char* wh = getenv("SGX_TRK_TEST_WITHHOLD");
            const int nba = (wh && wh[0] == '1' && K.split > 1) ? n_blocks - 8 : n_blocks;
            sgx_trk_any_launch(nba, st, r->d, c->d_codes, d_ch, d_out, d_done, K, d_prof, d_xch, d_err);
            c->timing.track_kernel = 6.f;
        }
        else if (use_tp) {   // (one lane per prompt chip)
            sgx_trk_tp_launch(n_blocks, st, r->d, c->d_codes, d_ch, d_out, d_done, K, d_prof, d_xch, d_err);
            c->timing.track_kernel = 3.f;
        } else {
            const char* wh = getenv("SGX_TRK_TEST_WITHHOLD");   // (the same test hook for the low-rate cooperative kernel)
            const int nb1 = (wh && wh[0] == '1' && K.split > 1) ? n_blocks - 8 : n_blocks;
            sgx_trk_multi_launch(nb1, st, r->d, c->d_codes, d_ch, d_out, d_done, K, d_prof, d_xch, d_err);
            c->timing.track_kernel = 4.f;
        }
        hipEventRecord(c->ev[4], st);
        stamp("kernel queued");
        c->timing.track_members = (float)members_now;
        c->timing.track_streamed = streaming ? 1.f : 0.f;
        e = hipGetLastError();
        int h_err2[2] = {0, 0};   // [0] flags | 1 + channel of a timeout; [1] 1 + channel of a block beyond the units
        if (fast_look) {
            const unsigned long long seq = ++c->trk_seq;
            if (e == hipSuccess) {
                trk_finish_kernel<<<1, SGX_TRK_LOOK_CH, 0, st>>>(d_err, d_done, n_ch, (TrkLook*)((char*)c->d_look + SGX_TRK_LOOK_OFFSET), seq);
                e = hipGetLastError();
            }
            stamp("finish kernel queued");
            if (e == hipSuccess) {
                // spin (the kernel takes tens of milliseconds), then sleep in the stream synchronisation
                const auto t0 = std::chrono::steady_clock::now();
                bool seen = false;
                for (unsigned sp = 0; !seen; ++sp) {
                    if (__atomic_load_n(&h_look->seq, __ATOMIC_ACQUIRE) == seq) seen = true;
                    else if ((sp & 4095u) == 4095u &&
                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;
                }
                if (!seen) {
                    e = hipStreamSynchronize(st);
                    if (e == hipSuccess && __atomic_load_n(&h_look->seq, __ATOMIC_ACQUIRE) != seq) {
                        sgx_set_error("tracking: the launch's result words were not written");
                        return SGX_E_HIP;
                    }
                }
                h_err2[0] = h_look->err[0];
                h_err2[1] = h_look->err[1];
            }
            stamp("result words seen");
        } else {
            if (e == hipSuccess) e = hipMemcpyAsync(h_err2, d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, st);
            stamp("error word copy queued");
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            stamp("stream synchronised");
        }
        h_err = h_err2[0];
        reserved.drop();
        // test hooks: treat the first launch as timed out ('1'), or the one that follows a stalled stream ('2'); treat
        // the first launch as a stalled stream (SGX_TRK_TEST_STALL=1)
        const char* th = getenv("SGX_TRK_TEST_TIMEOUT");
        const char* tst = getenv("SGX_TRK_TEST_STALL");
        if (e == hipSuccess && th && K.split > 1 && !fallback_one &&
            ((th[0] == '1' && launches == 0) || (th[0] == '2' && stream_tried)))
            h_err = 1;
        if (e == hipSuccess && tst && tst[0] == '1' && launches == 0) h_err = TRK_ERR_STREAM;
        if (e == hipSuccess && (h_err & TRK_ERR_STREAM) && !stream_tried && !fallback_one) {
            // the streaming record's watermark stalled (the copy stream could not run beside the kernel): repeat
            // with the same decomposition once the whole record is resident
            fprintf(stderr, "[sgx] tracking: the record did not stream in beside the kernel; repeating the launch "
                            "on the resident record\n");
            stream_tried = true;
            continue;
        }
        h_err &= ~TRK_ERR_STREAM;
        if (e == hipSuccess && (h_err & TRK_ERR_SCALE) && !v3_off) {
            // samples beyond what the speculative kernel's 2^30 fixed point holds in 48 bits (a record that clips all the
            // time): the round-3 kernel, whose 2^28 holds full-scale samples that all line up, tracks it
            fprintf(stderr, "[sgx] tracking: samples too strong for the speculative kernel's fixed point (2 048 samples add up "
                            "to 131 072 or more in magnitude); repeating the launch with the round-3 kernel\n");
            v3_off = true;
            continue;
        }
        h_err &= ~TRK_ERR_SCALE;
        if (e == hipSuccess && (h_err & TRK_ERR_RANGE) == 0) h_err &= 0xFFFF;
        if (e == hipSuccess && (h_err & TRK_ERR_RANGE)) {
            sgx_set_error("tracking: channel %d reached a block longer than the %d units of %d samples the kernel "
                          "provides (the code NCO left its plausible range; dllNoiseBandwidth %g)",
                          (h_err2[1] ? h_err2[1] : (h_err & 0xFFFF)) - 1, K.n_units, TRK_UNIT, S.dllNoiseBandwidth);
            return SGX_E_RANGE;
        }
        if (e != hipSuccess || h_err == 0 || K.split == 1) break;
        fprintf(stderr, "[sgx] tracking: channel %d timed out waiting for a cooperating workgroup (%d workgroups per "
                        "channel, are the CUs shared?); repeating the launch with one workgroup per channel\n", h_err - 1,
                members_now);
        fallback_one = true;
    }
    if (fast_look && e == hipSuccess) {
        for (int i = 0; i < n_ch; ++i) ms_done[i] = h_look->done[i];
        e = hipEventSynchronize(c->ev[4]);   // (the word is stored a moment before the kernels retire: the times need the event)
    } else {
        if (e == hipSuccess && !direct) e = hipMemcpyAsync(out, d_out, elems * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(ms_done, d_done, sizeof(int) * (size_t)n_ch, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    stamp("ms_done copied");
    if (want_prof && e == hipSuccess) {
        std::vector<long long> hp(T2_PROF_STRIDE * (size_t)n_ch);
        hipMemcpy(hp.data(), d_prof, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost);
        if (used_v2) {
            for (int i = 0; i < n_ch && i < 8; ++i)
                for (int mm = 0; mm < used_members; mm += (i == 0 ? 1 : used_members - 1))
                    fprintf(stderr, "[sgx trk2 profile] ch %d member %2d cycles/block: release->publish %.0f  publish->sums %.0f  "
                                    "sums->release %.0f\n", i, mm, (double)hp[T2_PROF_STRIDE * i + mm] / ms,
                            (double)hp[T2_PROF_STRIDE * i + 64 + mm] / ms, (double)hp[T2_PROF_STRIDE * i + 128 + mm] / ms);
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
    if (direct) {
        // entries never reached keep the reference's initial values (tracking.py:65-94): zeros or +Inf
        const StepLook* slook = (const StepLook*)((const char*)c->h_look + SGX_STEP_LOOK_OFFSET);   // (chained: preRun's table)
        for (int i = 0; i < n_ch; ++i) {
            if (chained && slook->prn[i] == 0) continue;      // (a channel that is off: the caller gets the first n_active rows)
            const int dn = chained ? ms_done[i] : ((ch[i].prn == 0) ? 0 : ms_done[i]);
            if (dn >= ms) continue;
            for (int sidx = 0; sidx < SGX_NUM_SERIES; ++sidx) {
                const bool zero = (sidx == 0) || (sidx >= 3 && sidx <= 8);
                double* row = out + ((size_t)i * SGX_NUM_SERIES + (size_t)sidx) * (size_t)ms;
                for (int t = dn; t < ms; ++t) row[t] = zero ? 0.0 : INFINITY;
            }
        }
    }
    if (h_err != 0) {
        sgx_set_error("tracking kernel: channel %d reported a timeout with split %d", h_err - 1, K.split);
        return SGX_E_HIP;
    }
    {
        const int rq = r->loader ? r->load_rc.load() : SGX_OK;   // the loader failed while the kernel ran
        if (rq != SGX_OK) return sgx_if_require(r, r->n);
    }
    hipEventElapsedTime(&c->timing.track_ms, c->ev[3], c->ev[4]);
    stamp("done");
    if (floaty && K.fscale != 1.0) {
        // the kernel tracked fscale x the record: the six correlator series carry the factor (a power of two: exact),
        // everything the discriminators made of them (ratios) does not
        const double un = 1.0 / K.fscale;
        for (int i = 0; i < n_ch && !chained; ++i) {
            if (ch[i].prn == 0) continue;
            double* o = out + (size_t)i * SGX_NUM_SERIES * (size_t)ms;
            const int dn = ms_done[i] < ms ? ms_done[i] : ms;
            for (int sidx = 3; sidx <= 8; ++sidx)
                for (int t = 0; t < dn; ++t) o[(size_t)sidx * ms + t] *= un;
        }
    }
    return SGX_OK;
}

int sgx_track_kind(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch,
                   int32_t ms, double* out, int32_t* ms_done, int kind, long long skip_bytes, double fscale) {
    return track_kind_impl(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done, kind, skip_bytes, fscale, false);
}

// include/sgx.h: preRun on the device behind the pending acquisition, the tracking kernel behind it, ONE wait.
extern "C" int sgx_track_chained(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, int32_t n_ch, int32_t ms, double* out,
                                 int32_t* ms_done, int32_t data_type, int32_t* prn, double* acquiredFreq, double* codePhase,
                                 int32_t* n_active) {
    SGX_CHECK_ARG(c && r && out && ms_done && prn && acquiredFreq && codePhase && n_active);
    // (n_ch <= 8: the eager sequence launches the ACTIVE channels only, and which kernel runs depends on their number
    // rounded up to 8 - with at most 8 configured channels that is the same launch whatever preRun finds)
    if (c->acq_pending.mode != 1 || n_ch < 1 || n_ch > 8) return SGX_E_DEFER;
    if (data_type != SGX_DT_INT8 && data_type != SGX_DT_UINT8) return SGX_E_DEFER;
    const int rc = track_kind_impl(c, r, rec_file_offset, nullptr, n_ch, ms, out, ms_done, data_type,
                                   (long long)c->s.skipNumberOfBytes, 0.0, true);
    if (rc != SGX_OK) return rc;
    // (the stream has been synchronised: the page is complete)
    const StepLook* look = (const StepLook*)((const char*)c->h_look + SGX_STEP_LOOK_OFFSET);
    if (look->n_ch != n_ch || look->flags != 0) return SGX_E_DEFER;   // a NaN metric, a failed search, a channel in front of
                                                                       // the record: nothing was tracked, the eager calls report it
    *n_active = look->n_active;
    for (int i = 0; i < n_ch; ++i) {
        prn[i] = look->prn[i];
        acquiredFreq[i] = look->acquiredFreq[i];
        codePhase[i] = look->codePhase[i];
    }
    return SGX_OK;
}

extern "C" int sgx_track(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch,
                         int32_t n_ch, int32_t ms, double* out, int32_t* ms_done) {
    SGX_CHECK_ARG(c);
    return sgx_track_kind(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done, SGX_DT_INT8, (long long)c->s.skipNumberOfBytes, 0.0);
}

extern "C" int sgx_track_ex(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch,
                            int32_t n_ch, int32_t ms, double* out, int32_t* ms_done, int32_t data_type) {
    SGX_CHECK_ARG(c);
    if (data_type == SGX_DT_FLOAT32) return sgx_track_float32(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done);
    if (data_type == SGX_DT_FLOAT64) return sgx_track_float64(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done);
    if (sgx_dt_bytes(data_type) == 0) {
        sgx_set_error("sgx_track_ex: data_type %d is not one of SGX_DT_* (include/sgx.h)", (int)data_type);
        return SGX_E_ARG;
    }
    return sgx_track_kind(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done, data_type, (long long)c->s.skipNumberOfBytes, 0.0);
}
