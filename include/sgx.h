/* sgx.h - C-ABI of libsgx.so: MI355X (gfx950) GPS L1 C/A acquisition + tracking engine.
 *
 * The reference (perrysou/SoftGNSS-python) has no FFI; its boundary is the Python object API
 *   Settings                              reference initialize.py:80-185
 *   AcquisitionResult.acquire(longSignal) reference acquisition.py:27-204
 *   AcquisitionResult.preRun()            reference acquisition.py:259-306  (host glue, stays in Python)
 *   TrackingResult.track(fid)             reference tracking.py:13-295
 * The entry points below sit directly under those methods: the drop-in Python modules in
 * softgnss-python_amd/ bind them with ctypes (INTEGRATION.md shows the stub).  Further down: the stages either
 * side of that path - Settings.probeData statistics, the navigation chain of
 * postNavigation.py / ephemeris.py / geoFunctions - each citing the reference lines it replaces.
 *
 * Conventions: plain C, int status return (0 = SGX_OK, <0 = error; text via sgx_last_error),
 * no exceptions/callbacks across the boundary, the CALLER owns every host buffer (the library
 * copies and never keeps a host pointer after return), opaque handles for the device context
 * and for IF records resident in HBM.  One context per device; a context is not thread-safe,
 * different contexts are.  No torch / framework types anywhere in the signatures.
 */
#ifndef SGX_H
#define SGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_OK          0
#define SGX_E_ARG      -1   /* bad argument */
#define SGX_E_HIP      -2   /* HIP runtime error (no device, launch failure, ...) */
#define SGX_E_NOMEM    -3   /* allocation failed */
#define SGX_E_INDEX    -4   /* the reference's IndexError: coarse code phase == samples-per-chip
                               (acquisition.py:152-153 builds index N) */
#define SGX_E_RCCL     -5   /* RCCL error / library not loadable */
#define SGX_E_RANGE    -6   /* record too short for the request; also where the reference's numpy code raises on
                               out-of-range data (the error text then starts with the exception's name) */
#define SGX_E_DEFER    -7   /* sgx_track_chained: the queued (deferred) sequence does not apply to this call - no
                               acquisition pending, a kernel other than the cooperative int8 / uint8 ones, a streaming
                               record, a launch that had to be repeated; the caller runs sgx_acquire_end, preRun and
                               sgx_track_ex instead (nothing has been lost: the search's results are still pending) */

#define SGX_NUM_SERIES 13   /* per-ms tracking series, in this order (tracking.py:255-275):
                               absoluteSample codeFreq carrFreq I_P I_E I_L Q_E Q_P Q_L
                               dllDiscr dllDiscrFilt pllDiscr pllDiscrFilt */
#define SGX_MAX_SATS   16

typedef struct sgx_ctx sgx_ctx;     /* device context: one per GPU */
typedef struct sgx_if sgx_if;       /* int8 IF record resident in HBM */
typedef struct sgx_comm sgx_comm;   /* RCCL communicator for the acquisition peak gather */

/* POD mirror of the reference's Settings attributes used on the path (initialize.py:85-173). */
typedef struct sgx_settings {
    double samplingFreq;          /* initialize.py:107 */
    double IF;                    /* initialize.py:105 */
    double codeFreqBasis;         /* initialize.py:109 */
    double acqSearchBand;         /* kHz, initialize.py:123 */
    double acqThreshold;          /* initialize.py:126 */
    double dllDampingRatio;       /* initialize.py:130 */
    double dllNoiseBandwidth;     /* initialize.py:132 */
    double dllCorrelatorSpacing;  /* initialize.py:134 */
    double pllDampingRatio;       /* initialize.py:137 */
    double pllNoiseBandwidth;     /* initialize.py:139 */
    int64_t skipNumberOfBytes;    /* initialize.py:94 */
    int32_t codeLength;           /* initialize.py:112 */
    int32_t numberOfChannels;     /* initialize.py:88 */
} sgx_settings;

/* One tracking channel as preRun() hands it over (acquisition.py:285-303). */
typedef struct sgx_chan_init {
    double acquiredFreq;
    double codePhase;   /* samples; tracking seeks to skipNumberOfBytes + codePhase (tracking.py:107) */
    int32_t prn;        /* 1-based; 0 = channel off */
    int32_t reserved;
} sgx_chan_init;

/* Integer-only synthetic scene (softgnss-python_amd/synth.py); host and device generators are
 * bit-identical. */
typedef struct sgx_sat {
    uint64_t code_fcw;  /* 32.32 chips per sample */
    uint64_t code_c0;   /* 32.32 code phase at sample 0 (whole code periods shift the navigation bit edges) */
    uint64_t nav_seed;
    uint32_t car_fcw;   /* carrier NCO word, cycles per sample * 2^32 */
    uint32_t car_ph0;
    int32_t prn;        /* 1-based */
    int32_t amp;
} sgx_sat;

typedef struct sgx_scene {
    uint64_t seed;
    int32_t n_sats;
    int32_t nav_mode;                      /* 0: hash navigation bits; 1: nav_bits tables (2048 bits per satellite) */
    sgx_sat sats[SGX_MAX_SATS];
    int16_t cos_lut[256];
    uint8_t nav_bits[SGX_MAX_SATS][256];   /* bit b of satellite s = (nav_bits[s][b >> 3] >> (b & 7)) & 1 */
} sgx_scene;

/* Timings of the last call, measured with HIP events on the context's stream. */
typedef struct sgx_timing {
    float acquire_ms;        /* whole sgx_acquire device time */
    float acq_coarse_ms;     /* mix + FFTs + correlation + peak search; the split is measured with SGX_ACQ_SPLIT_EVENT=1 */
    float acq_fine_ms;       /* fine-frequency FFTs                     (else: coarse = the whole call, fine = 0)          */
    float track_ms;          /* the tracking kernel */
    float synth_ms;          /* the generator kernel */
    float track_kernel;      /* which tracking kernel the last sgx_track ran: 2 latency-mode (sgx_trk2.hip), 3 throughput-mode
                                (sgx_trk_tp.hip), 4 low-rate (sgx_trk_multi.hip), 5 speculative latency-mode (sgx_trk3.hip),
                                6 per-sample, any sample type (sgx_trk_any.hip) */
    float track_members;     /* workgroups per channel of that launch */
    float track_streamed;    /* 1: the kernel followed the watermark of a record that was still streaming in */
} sgx_timing;

/* ---- library ------------------------------------------------------------------------------ */
const char* sgx_version(void);
int sgx_last_error(char* buf, size_t n);         /* copies the calling thread's last error text */

/* ---- host-side exact helpers (no GPU needed) ---------------------------------------------- */
/* samplesPerCode property, initialize.py:183-185 */
int sgx_samples_per_code(const sgx_settings* s, int64_t* n);
/* Settings.generateCAcode(prn0), initialize.py:234-302: out[1023] of +-1.0, prn0 in 0..31 */
int sgx_generate_ca_code(int32_t prn0, double* out);
/* Settings.makeCaTable(), initialize.py:188-231: out[32 * samplesPerCode] of +-1.0 */
int sgx_make_ca_table(const sgx_settings* s, double* out);
/* Settings.calcLoopCoef(LBW, zeta, k), initialize.py:304-328 */
int sgx_calc_loop_coef(double lbw, double zeta, double k, double* tau1, double* tau2);
/* Host evaluation of the short-chain arithmetic the tracking kernel's loop-filter waves use (csrc/sgx_trk_math.h; on the
 * host the hardware reciprocal seeds are replaced by float-precision ones).  Diagnostics for the parity tests:
 * fn 0: 1/a   1: a/b   2: sqrt(a)   3: atan(a/b)   4: out[0..1] = sin, cos of 2 pi a   5: ceil(a/b) */
int sgx_trk_math_eval(int32_t fn, double a, double b, double* out);

/* ---- device context and IF records --------------------------------------------------------- */
int sgx_device_count(int* n);
int sgx_ctx_create(const sgx_settings* s, int device, sgx_ctx** out);
/* The same with a stream priority class (-1 high, 0 normal, +1 low).  HIP multiplexes the streams of one priority
 * onto a few hardware queues, where two persistent tracking kernels would run one after the other; contexts that
 * are meant to run AT THE SAME TIME on one GPU (independent records) therefore take different classes. */
int sgx_ctx_create_prio(const sgx_settings* s, int device, int priority, sgx_ctx** out);
int sgx_ctx_destroy(sgx_ctx* c);
int sgx_ctx_sync(sgx_ctx* c);                     /* hipStreamSynchronize on the context stream */
int sgx_get_timing(sgx_ctx* c, sgx_timing* out);

/* Optional pinned host memory for the big result buffer of sgx_track (a plain pageable buffer works too,
 * it only copies slower). */
int sgx_host_alloc(size_t bytes, void** out);
int sgx_host_free(void* p);

/* np.fromfile(fid, 'int8', n) replacement: copy n host samples into a new HBM record
 * (initialize.py:481, tracking.py:154). */
int sgx_if_upload(sgx_ctx* c, const int8_t* host, size_t n, sgx_if** out);
/* The same straight from the record file (the reference's fid, initialize.py:466-481 / tracking.py:107,154):
 * bytes [file_offset, file_offset + n) of `path` (raw headerless int8 samples) are streamed through two
 * pinned staging buffers, the next pread overlapping the previous chunk's H2D copy.  A file shorter than
 * requested yields a shorter record (tracking then reports the reference's short-read exit). */
int sgx_if_upload_file(sgx_ctx* c, const char* path, uint64_t file_offset, size_t n, sgx_if** out);
/* The same record, but the call returns at once: a background thread streams the file into HBM in file order and
 * sgx_acquire / sgx_track / sgx_if_download / sgx_probe_stats wait for exactly the samples they need - sgx_track's
 * cooperative kernel follows a device-side watermark block by block, so tracking overlaps the transfer.
 * sgx_if_wait blocks until the first n samples (0 = all) are resident and reports a read error if there was one;
 * sgx_if_free joins the thread. */
int sgx_if_open_file(sgx_ctx* c, const char* path, uint64_t file_offset, size_t n, sgx_if** out);
int sgx_if_wait(sgx_ctx* c, sgx_if* r, size_t n);
/* Generate samples [offset, offset+n) of a synthetic scene directly in HBM. */
int sgx_if_synth(sgx_ctx* c, const sgx_scene* scene, uint64_t offset, size_t n, sgx_if** out);
int sgx_if_download(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n, int8_t* host);
int sgx_if_length(const sgx_if* r, size_t* n);
int sgx_if_free(sgx_ctx* c, sgx_if* r);

/* ---- AcquisitionResult.acquire (acquisition.py:27-204) -------------------------------------
 * Searches PRN indices prn0[0..n_prn) (0-based) on samples [offset, offset+n_samples) of the
 * record.  n_blocks 1-ms blocks feed the coarse search (reference: 2); noncoh = 0 keeps the
 * reference rule (block with the larger maximum, acquisition.py:129-133), noncoh = 1 sums
 * |corr|^2 over the blocks (extension, BASELINE.json config 4).  n_samples is the length of the
 * reference's longSignal: its mean is the DC removed before the fine search (acquisition.py:59)
 * and codePhase + 10 ms must fit in it (acquisition.py:177).
 * Outputs, each [n_prn]: the three reference fields (acquisition.py:201-203) and, for parity
 * checks, frequencyBinIndex and fftMaxIndex (-1 where the PRN is not detected).
 * Returns SGX_E_INDEX where the reference raises IndexError (outputs up to that PRN are valid).
 */
int sgx_acquire(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n_samples,
                const int32_t* prn0, int32_t n_prn, int32_t n_blocks, int32_t noncoh,
                double* carrFreq, double* codePhase, double* peakMetric,
                int32_t* freqBin, int32_t* fineIdx);

/* The same for a signal that is not int8: acquisition.py:55-59 works on whatever real dtype numpy hands it (float
 * samples, int16 values, a record rescaled on the host).  `signal` = n_samples fp64 samples on the host; they are copied to
 * HBM and the kernels read them in place of the int8 record (same fp64 arithmetic, same outputs as sgx_acquire). */
int sgx_acquire_f64(sgx_ctx* c, const double* signal, size_t n_samples, const int32_t* prn0, int32_t n_prn,
                    int32_t n_blocks, int32_t noncoh, double* carrFreq, double* codePhase, double* peakMetric,
                    int32_t* freqBin, int32_t* fineIdx);

/* ---- the same path without host round trips between its stages (round 6) --------------------
 * The reference's caller looks at every stage's result before it starts the next (initialize.py:484-506: acquire, preRun,
 * TrackingResult.track).  A caller that wants the tracking results can queue all three and wait once:
 *   sgx_acquire_begin    sgx_acquire's arguments without the outputs: the search is queued (acquisition.py:27-204), the
 *                        call returns without looking at it.  One acquisition may be pending per context.
 *   sgx_track_chained    preRun (acquisition.py:259-306: stable descending sort of the 32 peak metrics, the first
 *                        min(n_ch, #detected) become channels) runs ON THE DEVICE behind the pending search and the tracking
 *                        kernel (tracking.py:13-295) behind it, reading the channel table where preRun left it; the call
 *                        waits once, for everything.  out = [n_ch][13][ms] (pinned memory from sgx_host_alloc), ms_done =
 *                        [n_ch]; the channel table as preRun made it comes back in prn / acquiredFreq / codePhase [n_ch]
 *                        (prn 0 = off; the first *n_active channels are on, in order of descending metric).  Returns
 *                        SGX_E_DEFER when the queued sequence does not apply (see the code's comment; n_ch > 8) - then nothing has
 *                        been tracked and the eager calls do the work; SGX_E_INDEX / SGX_E_RANGE where the SEARCH failed
 *                        the way the reference's acquire() raises (no channel was tracked).
 *   sgx_acquire_end      the pending search's outputs (sgx_acquire's), without waiting if sgx_track_chained has run.
 * Results are those of sgx_acquire + preRun + sgx_track_ex, bit for bit: the same kernels in the same order, and the
 * device-side preRun repeats the host's arithmetic (tests/test_gpu_parity.py). */
int sgx_acquire_begin(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n_samples,
                      const int32_t* prn0, int32_t n_prn, int32_t n_blocks, int32_t noncoh);
int sgx_acquire_end(sgx_ctx* c, double* carrFreq, double* codePhase, double* peakMetric,
                    int32_t* freqBin, int32_t* fineIdx);
int sgx_track_chained(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, int32_t n_ch, int32_t ms,
                      double* out, int32_t* ms_done, int32_t data_type,
                      int32_t* prn, double* acquiredFreq, double* codePhase, int32_t* n_active);

/* ---- TrackingResult.track (tracking.py:13-295) ----------------------------------------------
 * Tracks n_ch channels for `ms` code periods on the record.  rec_file_offset is the byte offset
 * in the reference's file of the record's first sample (0 when the whole file was uploaded):
 * a channel starts at file byte skipNumberOfBytes + codePhase (tracking.py:107) and
 * absoluteSample is reported as a file position (tracking.py:255).
 * out is [n_ch][SGX_NUM_SERIES][ms] float64, pre-filled exactly like tracking.py:65-94 (0 or
 * +Inf) for entries never reached; ms_done[ch] = blocks completed (== ms unless the record ran
 * out, the reference's short-read exit tracking.py:159-163; channels with prn == 0 report 0).
 */
int sgx_track(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset,
              const sgx_chan_init* ch, int32_t n_ch, int32_t ms,
              double* out, int32_t* ms_done);

/* The same for a record of Settings.dataType samples (settings.dataType, initialize.py:60; read with
 * np.fromfile(fid, dataType, blksize) at tracking.py:154).  data_type SGX_DT_INT8 is sgx_track; otherwise the record
 * handle holds the file's BYTES as they are (little endian) and rec_file_offset, skipNumberOfBytes + codePhase and
 * absoluteSample stay BYTE positions, exactly as the reference's fid.seek / fid.tell treat them (tracking.py:107, 255) -
 * so a channel whose start byte lies inside a sample reads samples that straddle the file's, as it does there.
 * int8, uint8 and int16 have typed kernels (int16 and uint8 at samplingFreq >= 16 x codeFreqBasis).  SGX_DT_FLOAT32 is
 * tracked through them EXACTLY when every sample of the window is m 2^-k for one k and 16-bit integers m (floats written
 * from ADC samples, or normalised by a power of two) and the channels start on samples: the integers go through the int8
 * / int16 kernels and the correlator series are scaled back, which no rounding of the reference's float64 arithmetic can
 * tell from the real thing.  Other float32 records, and SGX_DT_FLOAT64, run the latency-mode kernel on samples scaled by a
 * power of two (the window is scanned once for its largest |x|; exact both ways).  Everything else - float16, the wider
 * integers, int16 / uint8 at low sampling rates, float records with NaN / infinite samples or channels that start inside
 * a sample - is read sample by sample where it lies and promoted to float64 as numpy promotes it (the per-sample kernel,
 * sgx_trk_any.hip: slower, same contract).  Complex types are not tracked (the reference's
 * discriminators fail on them). */
#define SGX_DT_INT8  0
#define SGX_DT_INT16 1
#define SGX_DT_UINT8 2   /* offset-binary bytes as the reference reads them with dataType 'uint8': no offset is removed */
#define SGX_DT_FLOAT32 3 /* IEEE binary32 */
#define SGX_DT_FLOAT64 4
#define SGX_DT_UINT16  5
#define SGX_DT_INT32   6
#define SGX_DT_UINT32  7
#define SGX_DT_INT64   8
#define SGX_DT_UINT64  9
#define SGX_DT_FLOAT16 10
int sgx_track_ex(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset,
                 const sgx_chan_init* ch, int32_t n_ch, int32_t ms,
                 double* out, int32_t* ms_done, int32_t data_type);

/* Which tracking kernel sgx_track_ex runs, and with how many cooperating workgroups (members) per channel - the one rule
 * the host applies (csrc/sgx_trk.hip: sgx_track_plan), without the diagnostic SGX_TRK_* overrides.  No reference
 * counterpart (tracking.py:59 is one serial loop); needs no GPU.  n_cus: compute units of the device (256 on an MI355X);
 * float_in_range != 0: a float32 / float64 record whose window was scanned and can run the typed kernel.
 *   kernel  2 trk2_kernel (latency mode, any member layout)   3 trk_kernel_tp (throughput mode, > 128 channels)
 *           4 trk_kernel_multi (< ~15.4 samples per chip)      5 trk3_kernel (speculative latency mode, the headline)
 *           6 trk_kernel_any (any sample type, sample by sample)
 *   members workgroups per channel (trk2_kernel with one workgroup per unit AND correlator arm: 3 x units) */
int sgx_track_plan(const sgx_settings* s, int32_t data_type, int32_t n_ch, int32_t n_cus, int32_t float_in_range,
                   int32_t* kernel, int32_t* members);

/* How sgx_acquire cuts the correlation batch of a call into chunks over its (one or two) queues - the host's one rule
 * (csrc/sgx_acq.hip: acq_plan), without the SGX_ACQ_* overrides; no reference counterpart (acquisition.py:93-133 is one
 * loop over PRNs and bins); needs no GPU.  chunk_rows <= 0: the default (348).  A chunk is prn_chunk whole PRNs
 * (bin_runs == 1) or one PRN's rows of bins_per_run Doppler bins (bin_runs > 1, non-coherent sums only). */
int sgx_acquire_plan(int32_t n_prn, int32_t n_bins, int32_t n_blocks, int32_t noncoh, int32_t chunk_rows,
                     int32_t max_queues, int32_t* prn_chunk, int32_t* bin_runs, int32_t* bins_per_run, int32_t* queues);
/* The two constants behind that rule: the default chunk size (rows; SGX_ACQ_CHUNK_ROWS overrides it) and the largest batch
 * of rows one launch takes. */
int sgx_acquire_plan_limits(int32_t* default_chunk_rows, int32_t* max_rows);

/* Measured HBM rates of this device for the roofline report (no reference counterpart): a read-only stream and a
 * copy (read + write bytes counted) over `bytes` of device memory, `reps` timed launches each, GB/s. */
int sgx_stream_rates(sgx_ctx* c, size_t bytes, int reps, double* read_gbs, double* copy_gbs);

/* ---- next row: raw-data statistics of Settings.probeData (initialize.py:330-417) ------------------------------
 * Window [offset, offset+n) of a resident record (the reference reads 10 * samplesPerCode samples,
 * initialize.py:369-371).  f[8193] (MHz) and pxx[8193] = welch(data - mean(data), fs_mhz, hamming(16384, False),
 * 16384, 1024, 16384) (initialize.py:389-394); hist[255] = np.histogram(data, arange(-128, 128))[0]
 * (initialize.py:400, last bin closed); *n_segments = Welch segments averaged.  SGX_E_RANGE ("ValueError") for
 * fewer than 16384 samples. */
#define SGX_PROBE_BINS 8193
#define SGX_PROBE_HIST 255
int sgx_probe_stats(sgx_ctx* c, const sgx_if* rec, size_t offset, size_t n, double fs_mhz, double* f, double* pxx,
                    int64_t* hist, int32_t* n_segments);

/* ---- next row: bit sync + preamble search on the tracking output (postNavigation.py:443-631) ----------------
 * NavigationResult.findPreambles: I_P is [n_ch][ms] float64 (row i = i-th record of the tracking results);
 * firstSubFrame[ch] = ms index of the first verified TLM preamble, 0 = none (then the reference drops the
 * channel from its active list).  The sign correlation against the 160-ms preamble runs on the device, the
 * 6000-ms spacing test and the TLM/HOW parity checks (navPartyChk) on the host.  SGX_E_RANGE where the
 * reference's numpy code raises on a slice cut short by the record ends (candidate within 40 ms of the start
 * or 1200 ms of the end); sgx_last_error() then starts with "ValueError" or "IndexError" accordingly. */
int sgx_find_preambles(sgx_ctx* c, const double* I_P, int32_t n_ch, int32_t ms, int32_t search_start,
                       int32_t* firstSubFrame);
/* NavigationResult.navPartyChk (postNavigation.py:443-521): ndat32 = D29* D30* d1..d24 D25..D30 as +-1;
 * flips d1..d24 in place when D30* != 1 like the reference; status +1 / -1 (parity ok) or 0. */
int sgx_nav_parity_check(double* ndat32, int32_t* status);

/* The bit integration at the head of postNavigate (postNavigation.py:125-138): I_P[start-20 : start+30000] of one
 * channel summed in 20-ms columns (numpy's summation order), bit = sum > 0.  bits must hold 1501 entries;
 * *n_bits = 1501 for a full slice, fewer where Python's slice is clipped; SGX_E_RANGE ("ValueError") when the
 * clipped slice is not a multiple of 20 ms.  Host code: 30 020 additions per channel. */
int sgx_nav_bits(const double* I_P_row, int32_t ms, int32_t subFrameStart, uint8_t* bits, int32_t* n_bits);

/* ephemeris.ephemeris (ephemeris.py:60-195): bits = n_bits >= 1500 values 0/1 starting at the first bit of a
 * subframe, d30star = last bit of the word before; eph[27] in the reference's field order (weekNumber, accuracy,
 * health, T_GD, IODC, t_oc, a_f2, a_f1, a_f0, IODE_sf2, C_rs, deltan, M_0, C_uc, e, C_us, sqrtA, t_oe, C_ic,
 * omega_0, C_is, i_0, C_rc, omega, omegaDot, IODE_sf3, iDot), *tow in seconds.  Like the reference, parity is not
 * checked here.  SGX_E_ARG ("TypeError") for fewer than 1500 bits, SGX_E_RANGE ("UnboundLocalError") when
 * subframe 1, 2 or 3 is not among the five.  Host code. */
#define SGX_EPH_FIELDS 27
int sgx_ephemeris(const uint8_t* bits, int32_t n_bits, uint8_t d30star, double* eph, int64_t* tow);

/* NavigationResult.calculatePseudoranges (postNavigation.py:27-72): absoluteSample is [n_rows][ms] (row i = i-th
 * record of the tracking results), msOfTheSignal[numberOfChannels] the measurement point per channel, channelList
 * the channels to use; pseudoranges[numberOfChannels] in metres, +inf for channels not listed (NaN everywhere if
 * the list is empty, as in the reference).  SGX_E_RANGE ("IndexError") for a point outside the series.  Host code. */
int sgx_pseudoranges(const double* absoluteSample, int32_t n_rows, int32_t ms, const double* msOfTheSignal,
                     const int32_t* channelList, int32_t n_list, int32_t numberOfChannels, int64_t samplesPerCode,
                     double startOffset, double c_mps, double* pseudoranges);

/* ---- next row: satellite positions and the least-squares fix (geoFunctions/__init__.py); scalar host code -------
 * eph is [32][SGX_EPH_FIELDS] in sgx_ephemeris order, row PRN-1.  Angles in degrees where the reference's are. */
int sgx_check_t(double time, double* corrTime);                                  /* geoFunctions/__init__.py:745-771 */
int sgx_e_r_corr(double traveltime, const double* X_sat, double* X_sat_rot);     /* :491-523, 3-vectors */
int sgx_togeod(double a, double finv, double X, double Y, double Z, double* dphi, double* dlambda, double* h); /* :892-996 */
int sgx_topocent(const double* X, const double* dx, double* Az, double* El, double* D);                      /* :1003-1064 */
int sgx_tropo(double sinel, double hsta, double p, double tkel, double hum, double hp, double htkel, double hhum,
              double* ddr);                                                                                 /* :1071-1186 */
/* satpos (:779-885): satPositions [3][n] (row-major, one column per entry of prnList), satClkCorr [n] seconds */
int sgx_satpos(double transmitTime, const int32_t* prnList, int32_t n, const double* eph, double* satPositions,
               double* satClkCorr);
/* leastSquarePos (:636-739): satpos [3][n], obs [n] metres; pos[4] = X Y Z dt, el / az [n], dop[5] = G P H V T.
 * *rank_deficient = 1 where the reference gives up (matrix_rank(A) != 4) and returns a zero position. */
int sgx_least_square_pos(const double* satpos, const double* obs, int32_t n, double c_mps, int32_t useTropCorr,
                         double* pos, double* el, double* az, double* dop, int32_t* rank_deficient);
int sgx_cart2geo(double X, double Y, double Z, int32_t i, double* phi, double* lambda_, double* h);   /* :7-77 */
int sgx_find_utm_zone(double latitude, double longitude, int32_t* utmZone);                          /* :529-571 */
int sgx_cart2utm(double X, double Y, double Z, int32_t zone, double* E, double* N, double* U);      /* :176-372 */

/* The measurement-epoch loop of NavigationResult.postNavigate (postNavigation.py:150-290): for epoch m = 0 .. n_meas-1
 * the channels in use are those of `ready` whose elevation at the previous solved epoch was >= elevationMask (all of
 * `ready` at first); pseudoranges at millisecond subFrameStart[ch] + navSolPeriod m (sgx_pseudoranges), satellite
 * positions at transmitTime = tow + m navSolPeriod / 1000 (sgx_satpos), and with more than three channels the
 * least-squares fix, geodetic and UTM coordinates (sgx_least_square_pos, sgx_cart2geo, sgx_find_utm_zone,
 * sgx_cart2utm); otherwise the epoch is flagged in not_enough[m] and its fix is NaN, as the reference leaves it.
 * absoluteSample is [n_rows][ms] and prn_of_row [n_rows] (results row k serves channel k, as in the reference).
 * Outputs, prefilled here as the reference prefills them: chan_PRN (zeros) and chan_el / chan_az / chan_rawP /
 * chan_correctedP (NaN) are [numberOfChannels][64]; DOP [5][64] zeros; X Y Z dt latitude longitude height E N U
 * [64] NaN each, in this order in `sol` ([10][64]); *utmZone the zone of the last solved epoch.  n_meas > 64 is the
 * reference's IndexError (SGX_E_RANGE).  Host code. */
int sgx_post_navigate(const double* absoluteSample, int32_t n_rows, int32_t ms, const int32_t* prn_of_row,
                      const double* subFrameStart, const int32_t* ready, int32_t n_ready, int32_t numberOfChannels,
                      const double* eph, int64_t tow, int64_t samplesPerCode, double startOffset, double c_mps,
                      double navSolPeriod, double elevationMask, int32_t useTropCorr, int32_t n_meas,
                      double* chan_PRN, double* chan_el, double* chan_az, double* chan_rawP, double* chan_correctedP,
                      double* DOP, double* sol, int32_t* utmZone, int32_t* not_enough);

/* ---- RCCL peak gather (multi-GPU acquisition shard) ------------------
 * (SURVEY.md section 8(b) sketched single-process sgx_group_* entry points - one host thread driving every device
 * through ncclCommInitAll.  The build runs ONE PROCESS PER GPU instead, as the bench contract launches it, so the
 * collective side of the boundary is a per-rank communicator; sharding itself is a loop over PRN / channel index
 * ranges on the caller's side: softgnss-python_amd/shard.py.)
 * One process per GPU.  Rank 0 calls sgx_comm_unique_id and ships the 128 bytes to the other
 * ranks by any host channel; every rank then calls sgx_comm_create.  sgx_comm_allgather
 * all-gathers `bytes` bytes per rank (host buffers, staged through HBM, ncclAllGather on the
 * context stream over xGMI). */
int sgx_comm_unique_id(uint8_t id[128]);
int sgx_comm_create(sgx_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t id[128], sgx_comm** out);
int sgx_comm_allgather(sgx_comm* m, const void* send, void* recv, size_t bytes);

/* The sharded search as ONE call (round 6; BASELINE configs[3]: acquisition.py:92 is the loop that shards, :135-193 do not
 * shrink).  Rank `rank` of `world` searches its contiguous, balanced share of PRN indices 0 .. n_prn_total-1 (the partition
 * of softgnss-python_amd/shard.py: plan_shards); its peaks are packed into 40-byte records on the device behind the search,
 * ONE ncclAllGather on the context's stream gathers every rank's, a small kernel copies them to a pinned page and the host
 * looks once.  Outputs: the merged 32-entry arrays of acquisition.py:201-203 plus frequencyBinIndex / fftMaxIndex (-1 where
 * not detected), identical on every rank.  comm NULL: no collective - world 1, or one rank's shard run alone (what one
 * rank of an N-GPU run executes, for timing; only its own PRNs are filled in).  A rank whose search fails the way the
 * reference's acquire() raises (SGX_E_INDEX / SGX_E_RANGE) marks its record and EVERY rank returns that error. */
int sgx_acquire_sharded(sgx_ctx* c, sgx_comm* comm, int32_t rank, int32_t world, const sgx_if* r, size_t offset,
                        size_t n_samples, int32_t n_prn_total, int32_t n_blocks, int32_t noncoh,
                        double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin, int32_t* fineIdx);
int sgx_comm_destroy(sgx_comm* m);

#ifdef __cplusplus
}
#endif
#endif /* SGX_H */
