// Internal declarations shared by the translation units of libsgx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "sgx.h"

#define SGX_VERSION_STR "sgx 0.1 (gfx950)"

void sgx_set_error(const char* fmt, ...);

#define SGX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            sgx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                          __LINE__);                                                           \
            return SGX_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define SGX_CHECK_ARG(cond)                                                 \
    do {                                                                    \
        if (!(cond)) {                                                      \
            sgx_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return SGX_E_ARG;                                               \
        }                                                                   \
    } while (0)

typedef double2 cplx;   // complex128 as (re, im)

// The samples an acquisition reads: the int8 record (Settings.dataType 'int8'), or - acquire() handed any other real
// array (acquisition.py:55-59 works on whatever numpy dtype it gets) - an fp64 copy of it.
struct SgxSig {
    const int8_t* i8;
    const double* f64;
#ifdef __HIPCC__
    __device__ __forceinline__ double at(long long i) const { return f64 ? f64[i] : (double)i8[i]; }
#endif
};

// Twiddle tables for one FFT length: W_N^t = hi[t >> lo_bits] * lo[t & lo_mask]
struct FftPlan {
    int64_t n = 0;
    int lo_bits = 0;
    cplx* tw_hi = nullptr;   // device
    cplx* tw_lo = nullptr;   // device
    std::vector<int> radices;
};

struct sgx_if {
    int8_t* d = nullptr;   // device pointer; allocation is padded by SGX_IF_PAD zero bytes
    size_t n = 0;
    size_t cap = 0;        // bytes of the allocation behind d (>= n + SGX_IF_PAD)
    int device = 0;
    // background file -> HBM streaming (sgx_if_open_file): samples [0, host_mark) are resident
    std::thread* loader = nullptr;
    std::atomic<size_t> host_mark{0};        // bytes whose copy has completed, as the host knows it
    std::atomic<int> load_rc{0};             // SGX_OK while running / after success, an error code otherwise
    std::atomic<bool> load_done{false};
    std::atomic<long long> mag_max{-1};      // largest sum of magnitudes (bytes read as int8) over 17 consecutive 128-byte
                                             // blocks = a bound for every 2 048-byte window; -1: not scanned yet (sgx_trk.hip)
    unsigned long long* d_mark = nullptr;    // the same watermark in device memory, advanced in copy-stream order
    hipStream_t copy_stream = nullptr;
    char load_err[256] = {0};
};
// Block until samples [0, end) of a (possibly still streaming) record are resident; returns the loader's status.
int sgx_if_require(const sgx_if* r, size_t end);
// float32 records by exact narrowing to int8 / int16 (sgx_trk_f32.hip)
int sgx_track_float32(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch, int32_t ms,
                      double* out, int32_t* ms_done);
int sgx_track_float64(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch, int32_t ms,
                      double* out, int32_t* ms_done);
#define SGX_IF_PAD 256

// A DEFERRED acquisition (sgx_acquire_begin, round 6): every kernel of the search is queued, the host has not looked.
// mode 1: the device-led sequence is in flight (the result page's seq2 will equal `seq`); mode 2: the path could not be
// deferred, the search ran eagerly and its outputs wait in res_* for sgx_acquire_end.
struct AcqPending {
    int mode = 0;
    unsigned long long seq = 0;
    int n_prn = 0;
    int prn0[32];
    long long N = 0, npts = 0, fine_len = 0;
    size_t n_samples = 0;
    int rc = 0;                 // mode 2: the eager search's return code
    double res_carr[32], res_cph[32], res_met[32];
    int res_fb[32], res_fi[32];
};

// What the device-side preRun + a chained tracking launch leave in the upper half of the result page (offset 2048)
struct StepLook {
    int n_ch, n_active;         // channels of the table, channels that are on (acquisition.py:289)
    int flags;                  // 1 a NaN among the metrics (the host sorts); 2 IndexError / range error of the search (no
                                // channel is on); 4 a channel starts before the record
    int pad;
    int prn[32];
    double acquiredFreq[32], codePhase[32];
};
static_assert(sizeof(StepLook) <= 2048, "upper half of the result page");
#define SGX_STEP_LOOK_OFFSET 2048
// the result "page" is two pages: [0, 2048) the search's CoarseLook, [2048, 4096) StepLook, [4096, 8192) the gathered peak
// records of sgx_acquire_sharded behind the word the host spins on
#define SGX_GATHER_LOOK_OFFSET 4096
// ... and [6144, 8192): what a tracking launch leaves for the host's look (sgx_trk.hip: trk_finish_kernel) - the word, the
// two error words, ms_done of up to SGX_TRK_LOOK_CH channels
#define SGX_TRK_LOOK_OFFSET 6144
#define SGX_TRK_LOOK_CH 256
#define SGX_LOOK_BYTES 8192

struct sgx_ctx {
    sgx_settings s;
    AcqPending acq_pending;
    int device = 0;
    int priority = 0;            // stream priority class of the context: -1 high, 0 normal, +1 low
    hipStream_t stream = nullptr;
    hipStream_t acq_stream2 = nullptr;        // second queue of the correlation batch (created on first use)
    hipEvent_t acq_ev2[2] = {nullptr, nullptr};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    sgx_timing timing;
    int64_t n_code = 0;          // samplesPerCode
    int8_t* d_codes = nullptr;   // [32][1023] +-1
    // acquisition scratch (lazily sized)
    FftPlan plan_code;           // length samplesPerCode
    FftPlan plan_fine;           // length 8 * 2^ceil(log2(10 N))
    FftPlan plan_probe;          // length 16384 (Welch segments of sgx_probe_stats)
    cplx* d_fwd = nullptr;       // [n_blocks][n_bins][N] mixed-signal spectra
    cplx* d_codefd = nullptr;    // [32][N] code spectra
    cplx* d_work[2] = {nullptr, nullptr};   // ping-pong [rows][N]
    double* d_pow = nullptr;     // [rows][N] correlation power
    cplx* d_fine[2] = {nullptr, nullptr};
    double* d_sig64 = nullptr;   // fp64 copy of a non-int8 signal handed to sgx_acquire_f64
    size_t cap_sig64 = 0;
    size_t cap_fwd = 0, cap_code = 0, cap_w0 = 0, cap_w1 = 0, cap_pow = 0, cap_f0 = 0, cap_f1 = 0;   // bytes
    void* d_small = nullptr;     // small result area
    int acq_sum_phase = 0;       // acq_front_kernel: which of the two record-sum slots this call adds into ...
    bool acq_sum_clean[2] = {false, false};   // ... and whether a slot is known to hold zero (the other call's set-up zeroed it)
    void* h_small = nullptr;     // pinned mirror
    void* h_look = nullptr;      // coherent pinned page a kernel publishes the coarse search's outcome to (host spins on it)
    void* d_look = nullptr;      // its device address
    unsigned long long look_seq = 0;
    unsigned long long trk_seq = 0;
    // tracking
    double* d_trk_out = nullptr;
    size_t trk_out_elems = 0;
    void* d_trk_aux = nullptr;   // per-call device state of sgx_track (channels, done, exchange, err, profile)
    // One record allocation, streaming watermark and copy stream kept from the last sgx_if_free: a caller that opens a
    // record file per step (the reference's, initialize.py:466-506) would otherwise pay hipMalloc + hipFree of 1.4 GB and
    // a stream creation every time (milliseconds against a 50 ms step).
    int8_t* spare_d = nullptr;
    size_t spare_cap = 0;
    unsigned long long* spare_mark = nullptr;
    hipStream_t spare_copy_stream = nullptr;
    std::mutex spare_mu;
    size_t trk_aux_cap = 0;
    // pinned staging buffers of the file streamer, kept between calls (pinning 64 MiB costs ~15 ms)
    void* stage[2] = {nullptr, nullptr};
    std::atomic<bool> stage_busy{false};
};

// sgx_host.cpp
// Compute units claimed by this process's cooperative tracking launches (all contexts of a device): `want` CUs are
// granted (returned) only if they fit next to what is already running, else 0.
int sgx_cu_reserve(int device, int cus_total, int want);
void sgx_cu_release(int device, int n);
int sgx_host_ca_code(int prn0, int8_t* out /*1023*/);
int64_t sgx_host_samples_per_code(const sgx_settings* s);

// sgx_fft.hip
int sgx_fft_plan_create(FftPlan* p, int64_t n);
void sgx_fft_plan_destroy(FftPlan* p);
// Forward DFT of `rows` contiguous rows of length p->n. Result lands in *result (a or b).
int sgx_fft_forward(const FftPlan* p, cplx* a, cplx* b, int64_t rows, hipStream_t st, cplx** result,
                    int64_t nonzero_len);

// Optional fusion of the acquisition's pointwise kernels into the first / last radix pass.
struct FftFuse {
    const cplx* mul_x = nullptr;   // first pass input = conj(mul_x[bk]) * mul_f[prn]
    const cplx* mul_f = nullptr;
    const int2* row_map = nullptr; // device (bk, prn) per row, or null for the regular batch layout
    int rows_per_prn = 1;
    int prn_base = 0;
    double* pmax = nullptr;        // last pass: per-workgroup (max, first index) of |.|^2 * inv_n^2
    int* parg = nullptr;
    double inv_n = 0.0;
};
int sgx_fft_forward_fused(const FftPlan* p, cplx* a, cplx* b, int64_t rows, hipStream_t st, cplx** result,
                          int64_t nonzero_len, const FftFuse* fuse);
int sgx_fft_last_pass_blocks(const FftPlan* p);

// Four-step transform with LDS-resident sub-transforms (sgx_fft.hip), for the lengths it is instantiated for.
struct Fft4Fuse {
    const cplx* mul_x = nullptr;   // columns kernel input = conj(mul_x[b * n_phi + phi][(i + shift) mod n]) * mul_f[prn][i]
    const cplx* mul_f = nullptr;
    const int2* bin_map = nullptr; // device, per Doppler bin: (phi index, circular shift)
    const int2* row_map = nullptr; // device (block * n_bins + bin, prn) per row, or null for the regular batch layout
    int n_bins = 1, n_phi = 1, rows_per_prn = 1, prn_base = 0;
    int n_blocks = 1, blocks_fast = 0;   // regular batch rows ordered (prn, bin, block) instead of (prn, block, bin)
    double* pmax = nullptr;        // rows kernel: per-workgroup (max, first index) of |.|^2 * inv_n^2 ...
    int* parg = nullptr;
    double* pout = nullptr;        // ... or the powers themselves, [rows / sum_blocks][n]
    double inv_n = 0.0;
    int sum_blocks = 1;            // powers of this many consecutive rows are added before the reduction / store
    // ... or, per output row p, the maximum power over the index ranges [sec[32 + p], sec[64 + p]) and [sec[96 + p],
    // sec[128 + p]) folded into second_out[p] (integer atomic max on the bit pattern; rows with sec[p] < 0 are skipped):
    // the second-peak search (acquisition.py:162) without a stored row
    const int* sec = nullptr;
    double* second_out = nullptr;
    // ... or, per output row and residue k mod sgx_fft4_residues(): the maximum power, the maximum of the residue's other
    // powers and the maximum's first index ([rows / sum_blocks][residues] each): peak AND second peak from one pass
    double* t2_b1 = nullptr;
    double* t2_b2 = nullptr;
    int* t2_i1 = nullptr;
};
bool sgx_fft_fine_supported(int64_t npts);
int sgx_fft_fine_partials(void);
int sgx_fft_fine_search(const FftPlan* plan, SgxSig x, const int8_t* codes, const int* det_prn /* host, <= 32 */,
                        const int* d_det_phase, int n_det, long long len, const long long* d_sum, double n_mean, double ts,
                        double tc1, cplx* work, long long lo, long long hi, double* pv, long long* pi, hipStream_t st,
                        const int* d_det = nullptr /* device-led: [0] n_det, [1 + d] PRN index, [33 + d] code phase,
                                                      [80] arrival counter (zero) */,
                        long long* out_bi = nullptr /* device-led: [32] arg-max per detection (pinned page) ... */,
                        unsigned long long* out_seq = nullptr /* ... then this word = seq */, unsigned long long seq = 0,
                        const int* stage_src = nullptr /* device-led: dwords copied to stage_dst (the page) before the word */,
                        int* stage_dst = nullptr, int stage_words = 0);
bool sgx_fft4_supported(int64_t n);
int sgx_fft4_row_blocks(void);
int sgx_fft4_residues(void);
int sgx_fft4_forward(const FftPlan* p, const cplx* in, cplx* work, cplx* out, int64_t rows, hipStream_t st,
                     const Fft4Fuse* fuse);

// sgx_acq.hip: preRun (acquisition.py:259-306) on the device, behind a deferred acquisition: channel table -> d_ch (what the
// tracking kernels read) and the result page's StepLook.  d_ch: TrkChan[n_ch] in device memory.
struct TrkChan {      // one channel as the tracking kernels read it
    double acquiredFreq;
    long long pos0;   // record index of the channel's first sample
    int prn;          // 1-based, 0 = off
    int pad;          // two-byte samples: byte shift (0 / 1) of the channel's sample grid in the record; else 0
};
int sgx_prerun_enqueue(sgx_ctx* c, TrkChan* d_ch, int n_ch, long long skip_bytes, long long rec_file_offset, int sample_bytes);
// Wait for a deferred acquisition and decode it (the tail of the eager call); clears c->acq_pending.
int sgx_acquire_finish(sgx_ctx* c, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin, int32_t* fineIdx);

// sgx_host.cpp: the RCCL communicator of a context (librccl.so by dlopen)
struct sgx_comm {
    sgx_ctx* ctx;
    void* comm;
    int n_ranks, rank;
    void* d_send;
    void* d_recv;
    size_t cap;
};
int sgx_comm_allgather_device(sgx_comm* m, size_t bytes);

// sgx_synth.hip / sgx_acq.hip / sgx_trk.hip provide the C-ABI entry points directly.
