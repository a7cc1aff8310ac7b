// AcquisitionResult.acquire on gfx950 (reference acquisition.py:27-204).
//
// Data flow per call (everything fp64 / complex128, int8 IF read once):
//   mix      x[n]*(sin,cos)(f_k * phasePoints[n])        -> [blocks][bins][N]      (PRN independent,
//   FFT_N                                                 -> spectra X[b][k]         computed ONCE; the
//                                                                                    reference redoes it
//                                                                                    per PRN, Q6)
//   code     table[p][n] = ca[p][ceil((ts*k)/tc)-1]      -> FFT_N -> F[p]
//   corr     conj(X[b][k]) * F[p]  -> FFT_N  (= conj(N * ifft(X conj F)))  -> |.|^2 / N^2
//   peaks    per row max / first argmax (device), block choice + exclusion list (host, tiny),
//            second peak over the exclusion list (device)
//   fine     (x - mean) * code(floor((ts*k)/tc) mod 1023), zero-padded 2^22-point FFT, argmax of |X|
// HBM-resident scratch replaces the reference's per-PRN numpy temporaries.
#include <math.h>
#include <chrono>

#include "sgx_internal.h"

#define ACQ_MAX_BINS 128
#define ACQ_MAX_ROWS 2048
#define ACQ_DEFAULT_CHUNK_ROWS 348   // correlation rows per chunk (the intermediate then stays in the Infinity Cache)

struct MixArgs {
    double frq[ACQ_MAX_BINS];
    int n_bins;
    int n_blocks;
};

// acquisition.py:62-117: phasePoints[n] = ((n*2)*pi)*ts ; theta = frq*phasePoints ; I = sin*x, Q = cos*x
__global__ __launch_bounds__(256) void acq_mix_kernel(SgxSig x, cplx* __restrict__ out,
                                                      long long n, double ts, MixArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = blockIdx.y % a.n_bins;
    const int b = blockIdx.y / a.n_bins;
    const double pp = ((double)(i * 2) * M_PI) * ts;
    const double th = a.frq[k] * pp;
    double s, c;
    sincos(th, &s, &c);
    const double xv = x.at((long long)b * n + i);
    out[((long long)b * a.n_bins + k) * n + i] = make_double2(s * xv, c * xv);
}

// initialize.py:210-226 (A3) on the device, same IEEE operations: idx = ceil((ts*k)/tc) - 1
__global__ __launch_bounds__(256) void acq_code_kernel(const int8_t* __restrict__ codes,
                                                       const int* __restrict__ prn0, cplx* __restrict__ out,
                                                       long long n, double ts, double tc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = prn0[blockIdx.y];
    int idx = (int)ceil((ts * (double)(i + 1)) / tc) - 1;
    if (i == n - 1) idx = 1022;
    idx = idx < 0 ? 0 : (idx > 1022 ? 1022 : idx);
    out[(long long)blockIdx.y * n + i] = make_double2((double)codes[p * 1023 + idx], 0.0);
}

// rows r = (pi, b, k): Y = conj(X[b][k]) * F[pi]
__global__ __launch_bounds__(256) void acq_mul_kernel(const cplx* __restrict__ X, const cplx* __restrict__ F,
                                                      cplx* __restrict__ Y, long long n, int rows_per_prn,
                                                      int prn_base) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = blockIdx.y;
    const int pi = r / rows_per_prn;
    const int bk = r % rows_per_prn;
    const cplx xv = X[(long long)bk * n + i];
    const cplx fv = F[(long long)(prn_base + pi) * n + i];
    // conj(x) * f
    Y[(long long)r * n + i] =
        make_double2(__builtin_fma(xv.x, fv.x, xv.y * fv.y), __builtin_fma(xv.x, fv.y, -(xv.y * fv.x)));
}

// |fft|^2 / N^2 per output row (optionally summed over the blocks: noncoherent extension),
// plus row max and FIRST argmax (numpy argmax semantics, acquisition.py:139-143).
__global__ __launch_bounds__(256) void acq_power_kernel(const cplx* __restrict__ Z, double* __restrict__ P,
                                                        double* __restrict__ rowmax, int* __restrict__ rowarg,
                                                        long long n, double inv_n, int n_bins, int n_blocks,
                                                        int noncoh) {
    const int ro = blockIdx.x;   // output row
    double best = -1.0;
    int arg = 0;
    double* __restrict__ prow = P + (long long)ro * n;
    for (long long i = threadIdx.x; i < n; i += 256) {
        double v;
        if (noncoh) {
            const int pi = ro / n_bins, k = ro % n_bins;
            v = 0.0;
            for (int b = 0; b < n_blocks; ++b) {
                const cplx z = Z[(((long long)pi * n_blocks + b) * n_bins + k) * n + i];
                const double re = z.x * inv_n, im = z.y * inv_n;
                const double pw = re * re + im * im;
                v = (b == 0) ? pw : v + pw;
            }
        } else {
            const cplx z = Z[(long long)ro * n + i];
            const double re = z.x * inv_n, im = z.y * inv_n;
            v = re * re + im * im;
        }
        prow[i] = v;
        if (v > best) {
            best = v;
            arg = (int)i;
        }
    }
    __shared__ double s_v[256];
    __shared__ int s_i[256];
    s_v[threadIdx.x] = best;
    s_i[threadIdx.x] = arg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = s_v[threadIdx.x + s];
            const int oi = s_i[threadIdx.x + s];
            if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = ov;
                s_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rowmax[ro] = s_v[0];
        rowarg[ro] = s_i[0];
    }
}

// finish the fused last pass: per-workgroup (max, first index) partials -> one per row
__global__ __launch_bounds__(64) void acq_rowmax_finish_kernel(const double* __restrict__ pmax,
                                                               const int* __restrict__ parg, int nblk,
                                                               double* __restrict__ rowmax, int* __restrict__ rowarg) {
    const int row = blockIdx.x;
    double best = -1.0;
    int arg = 0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        const double v = pmax[(long long)row * nblk + b];
        const int i = parg[(long long)row * nblk + b];
        if (v > best || (v == best && i < arg)) {
            best = v;
            arg = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o);
        const int oi = __shfl_down(arg, o);
        if (ov > best || (ov == best && oi < arg)) {
            best = ov;
            arg = oi;
        }
    }
    if (threadIdx.x == 0) {
        rowmax[row] = best;
        rowarg[row] = arg;
    }
}

struct SecondArgs {
    int row[32];          // power row to search, -1 = skip
    int lo0[32], hi0[32]; // first index range [lo0, hi0)
    int lo1[32], hi1[32]; // second index range
};

// acquisition.py:162: max of the chosen frequency row over the exclusion index list
// (grid (PRNs, SEC_SPLIT): every workgroup takes a slice of the ranges and folds its maximum into out[p] with an integer
// atomic max on the bit pattern - powers are non-negative, so the patterns order like the values; out[] starts at 0)
#define SEC_SPLIT 16
__global__ __launch_bounds__(256) void acq_second_kernel(const double* __restrict__ P, double* __restrict__ out,
                                                         long long n, const SecondArgs* __restrict__ ap) {
    const SecondArgs& a = *ap;
    const int p = blockIdx.x;
    const int t0 = blockIdx.y * 256 + threadIdx.x, ts = gridDim.y * 256;
    double best = 0.0;
    if (a.row[p] >= 0) {
        const double* __restrict__ prow = P + (long long)a.row[p] * n;
        for (int i = a.lo0[p] + t0; i < a.hi0[p]; i += ts) best = fmax(best, prow[i]);
        for (int i = a.lo1[p] + t0; i < a.hi1[p]; i += ts) best = fmax(best, prow[i]);
    }
    __shared__ double s_v[256];
    s_v[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_v[threadIdx.x] = fmax(s_v[threadIdx.x], s_v[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((unsigned long long*)&out[p], (unsigned long long)__double_as_longlong(s_v[0]));
}

// the same on a recomputed complex correlation row: |z|^2 / N^2 formed on the fly, identical arithmetic to
// the fused last pass, so peak / second peak is a ratio of consistently rounded values
__global__ __launch_bounds__(256) void acq_second_cplx_kernel(const cplx* __restrict__ Z, double* __restrict__ out,
                                                              long long n, double inv_n, const SecondArgs* __restrict__ ap) {
    const SecondArgs& a = *ap;
    const int p = blockIdx.x;
    const int t0 = blockIdx.y * 256 + threadIdx.x, ts = gridDim.y * 256;
    double best = 0.0;
    if (a.row[p] >= 0) {
        const cplx* __restrict__ zrow = Z + (long long)a.row[p] * n;
        for (int i = a.lo0[p] + t0; i < a.hi0[p]; i += ts) {
            const double re = zrow[i].x * inv_n, im = zrow[i].y * inv_n;
            best = fmax(best, re * re + im * im);
        }
        for (int i = a.lo1[p] + t0; i < a.hi1[p]; i += ts) {
            const double re = zrow[i].x * inv_n, im = zrow[i].y * inv_n;
            best = fmax(best, re * re + im * im);
        }
    }
    __shared__ double s_v[256];
    s_v[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_v[threadIdx.x] = fmax(s_v[threadIdx.x], s_v[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((unsigned long long*)&out[p], (unsigned long long)__double_as_longlong(s_v[0]));
}

// integer sum of the record window (mean for acquisition.py:59): bytes up to the first 16-byte boundary, 16 bytes per
// lane from there, bytes again for the rest
__global__ __launch_bounds__(256) void acq_sum_kernel(const int8_t* __restrict__ x, long long n,
                                                      long long* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, gsz = (long long)gridDim.x * 256;
    long long head = (16 - ((unsigned long long)x & 15)) & 15;
    if (head > n) head = n;
    const long long n16 = (n - head) / 16;
    long long acc = 0;
    if (gid < head) acc += x[gid];
    const uint4* __restrict__ x16 = reinterpret_cast<const uint4*>(x + head);
    for (long long i = gid; i < n16; i += gsz) {
        const uint4 v = x16[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            acc += (int)(w[d] << 24) >> 24, acc += (int)(w[d] << 16) >> 24, acc += (int)(w[d] << 8) >> 24, acc += (int)w[d] >> 24;
    }
    for (long long i = head + n16 * 16 + gid; i < n; i += gsz) acc += x[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    // one atomic per workgroup (a thousand 64-bit atomics on one address took most of this kernel's time)
    __shared__ long long s_acc[4];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd((unsigned long long*)out, (unsigned long long)(s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3]));
}

// Call set-up in one launch: the PRN list and the bin map (by value) to their device tables, the accumulators zeroed.
struct AcqSetup {
    int prn[32];
    int2 bin[ACQ_MAX_BINS];
    int n_prn, n_bins;
};
__global__ __launch_bounds__(128) void acq_setup_kernel(AcqSetup a, int* __restrict__ d_prn, int2* __restrict__ d_bin,
                                                        long long* __restrict__ d_sum, double* __restrict__ d_second,
                                                        int* __restrict__ d_arrived = nullptr) {
    const int t = threadIdx.x;
    if (t < a.n_prn) d_prn[t] = a.prn[t];
    if (t < a.n_bins) d_bin[t] = a.bin[t];
    if (t < 32) d_second[t] = 0.0;
    if (t < 64 && d_arrived) d_arrived[t] = 0;   // [32] per PRN, [32] PRNs finished
    if (t == 0) d_sum[0] = 0;
}

// the same for an fp64 signal: one workgroup, fixed summation order (reproducible); the double's bits go to the same slot
__global__ __launch_bounds__(1024) void acq_sum_f64_kernel(const double* __restrict__ x, long long n,
                                                           long long* __restrict__ out) {
    __shared__ double s_v[1024];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) acc += x[i];
    s_v[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_v[threadIdx.x] += s_v[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = __double_as_longlong(s_v[0]);
}

// acquisition.py:170-177 (A9): xCarrier = (x - mean)[c : c+10N] * code[floor((ts*k)/tc1) mod 1023].
// Two detected PRNs share one complex row (first -> real part, second -> imaginary part): the two real-input
// spectra are separated again in the argmax kernel, which halves the 2^22-point FFT work.
__global__ __launch_bounds__(256) void acq_fine_prep_kernel(SgxSig x,
                                                            const int8_t* __restrict__ codes, cplx* __restrict__ out,
                                                            long long len, long long row_stride, double mean,
                                                            double ts, double tc1, const int* __restrict__ det_prn,
                                                            const int* __restrict__ det_phase, int n_det) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const int r = blockIdx.y;
    const double v = floor((ts * (double)(i + 1)) / tc1);
    const int chip = (int)((long long)v % 1023);
    const int d0 = 2 * r, d1 = 2 * r + 1;
    const double a = (x.at(det_phase[d0] + i) - mean) * (double)codes[det_prn[d0] * 1023 + chip];
    const double b = (d1 < n_det) ? (x.at(det_phase[d1] + i) - mean) * (double)codes[det_prn[d1] * 1023 + chip] : 0.0;
    out[(long long)r * row_stride + i] = make_double2(a, b);
}

// acquisition.py:182-187: argmax of |X_d[4 : uniq-5]| (first occurrence) for detection d, where the row holds
// Z = FFT(x_a + i x_b):  X_a[k] = (Z[k] + conj(Z[M-k]))/2,  X_b[k] = (Z[k] - conj(Z[M-k]))/(2i).
// Only the argmax is observable, so the common factor 1/4 of |.|^2 is dropped.
__global__ __launch_bounds__(256) void acq_fine_argmax_kernel(const cplx* __restrict__ X, long long row_stride,
                                                              long long lo, long long hi,
                                                              double* __restrict__ pv, long long* __restrict__ pi) {
    const int d = blockIdx.y;
    const cplx* __restrict__ row = X + (long long)(d >> 1) * row_stride;
    const double sgn = (d & 1) ? -1.0 : 1.0;
    double best = -1.0;
    long long arg = lo;
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
        const cplx z = row[i];
        const cplx w = row[row_stride - i];           // i >= 4 > 0, so M - i is inside the row
        const double re = z.x + sgn * w.x, im = z.y - sgn * w.y;   // z +- conj(w)
        const double v = re * re + im * im;
        if (v > best) {
            best = v;
            arg = i;
        }
    }
    __shared__ double s_v[256];
    __shared__ long long s_i[256];
    s_v[threadIdx.x] = best;
    s_i[threadIdx.x] = arg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = s_v[threadIdx.x + s];
            const long long oi = s_i[threadIdx.x + s];
            if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = ov;
                s_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pv[(long long)d * gridDim.x + blockIdx.x] = s_v[0];
        pi[(long long)d * gridDim.x + blockIdx.x] = s_i[0];
    }
}

// Forward rows of the shifted-spectrum search.  The reference transforms
// x (sin th + j cos th) = j x e^(-j th), th = 2 pi f n / fs, for every Doppler bin f (acquisition.py:103-117).  With
// f N / fs = s + phi (s integer, 0 <= phi < 1) that transform is j X_phi[(m + s) mod N], X_phi = fft(x e^(-j 2 pi phi n / N)):
// bins that share phi share ONE forward spectrum, read with a circular shift (|j| = 1 drops out of |.|^2).  For the
// default front end (fs / N = 1 kHz, 500 Hz grid) phi is 0 or 1/2: two forward transforms per 1-ms block instead of 29.
struct PhiArgs {
    double phi[4];
    int n_phi;
};
__global__ __launch_bounds__(256) void acq_mixphi_kernel(SgxSig x, cplx* __restrict__ out,
                                                         long long n, PhiArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = blockIdx.y % a.n_phi;
    const int b = blockIdx.y / a.n_phi;
    const double xv = x.at((long long)b * n + i);
    double s = 0.0, c = 1.0;
    if (a.phi[j] != 0.0) sincospi((2.0 * a.phi[j]) * ((double)i / (double)n), &s, &c);
    out[(long long)blockIdx.y * n + i] = make_double2(c * xv, -(s * xv));
}

// The front of a call in ONE launch (round 4; int8 records): the first kernels of a call are a few microseconds each and
// the host cannot queue them faster than they run, so four launches cost four launch latencies.  Workgroup roles by index:
// [0, n_mix) acq_mixphi_kernel's tiles, [n_mix, n_mix + n_code) acq_code_kernel's (PRN list by value), then ACQ_SUM_WGS
// of acq_sum_kernel's, then one of acq_setup_kernel's.  The record sum goes to the slot `sum_now`, which the PREVIOUS
// call's set-up workgroup zeroed (two slots alternate; the host keeps track and clears a slot itself when it cannot know).
#define ACQ_SUM_WGS 64
__global__ __launch_bounds__(256) void acq_front_kernel(AcqSetup su, SgxSig x, PhiArgs pa, const int8_t* __restrict__ codes,
                                                        cplx* __restrict__ out, long long n, int rows_fwd, double ts,
                                                        double tc, long long n_samples, int* __restrict__ d_prn,
                                                        int2* __restrict__ d_bin, long long* __restrict__ sum_now,
                                                        long long* __restrict__ sum_next, double* __restrict__ d_second,
                                                        int* __restrict__ d_arrived) {
    const int gx = (int)((n + 255) / 256);
    const int n_mix = rows_fwd * gx, n_code = su.n_prn * gx;
    int blk = blockIdx.x;
    const int t = threadIdx.x;
    if (blk < n_mix) {
        const int row = blk / gx;
        const long long i = (long long)(blk - row * gx) * 256 + t;
        if (i >= n) return;
        const int j = row % pa.n_phi;
        const int b = row / pa.n_phi;
        const double xv = x.at((long long)b * n + i);
        double s = 0.0, c = 1.0;
        if (pa.phi[j] != 0.0) sincospi((2.0 * pa.phi[j]) * ((double)i / (double)n), &s, &c);
        out[(long long)row * n + i] = make_double2(c * xv, -(s * xv));
        return;
    }
    blk -= n_mix;
    if (blk < n_code) {
        const int row = blk / gx;
        const long long i = (long long)(blk - row * gx) * 256 + t;
        if (i >= n) return;
        const int p = su.prn[row];
        int idx = (int)ceil((ts * (double)(i + 1)) / tc) - 1;
        if (i == n - 1) idx = 1022;
        idx = idx < 0 ? 0 : (idx > 1022 ? 1022 : idx);
        out[(long long)(rows_fwd + row) * n + i] = make_double2((double)codes[p * 1023 + idx], 0.0);
        return;
    }
    blk -= n_code;
    if (blk < ACQ_SUM_WGS) {
        const int8_t* __restrict__ xs = x.i8;
        const long long gid = (long long)blk * 256 + t, gsz = (long long)ACQ_SUM_WGS * 256;
        long long head = (16 - ((unsigned long long)xs & 15)) & 15;
        if (head > n_samples) head = n_samples;
        const long long n16 = (n_samples - head) / 16;
        long long acc = 0;
        if (gid < head) acc += xs[gid];
        const uint4* __restrict__ x16 = reinterpret_cast<const uint4*>(xs + head);
        for (long long i = gid; i < n16; i += gsz) {
            const uint4 v = x16[i];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
                acc += (int)(w[d] << 24) >> 24, acc += (int)(w[d] << 16) >> 24, acc += (int)(w[d] << 8) >> 24, acc += (int)w[d] >> 24;
        }
        for (long long i = head + n16 * 16 + gid; i < n_samples; i += gsz) acc += xs[i];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        __shared__ long long s_acc[4];
        if ((t & 63) == 0) s_acc[t >> 6] = acc;
        __syncthreads();
        if (t == 0) atomicAdd((unsigned long long*)sum_now, (unsigned long long)(s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3]));
        return;
    }
    if (t < su.n_prn) d_prn[t] = su.prn[t];
    if (t < su.n_bins) d_bin[t] = su.bin[t];
    if (t < 32) d_second[t] = 0.0;
    if (t < 64) d_arrived[t] = 0;   // [32] rows finished per PRN, [32] PRNs finished
    if (t == 0) sum_next[0] = 0;
}

static int ensure_buf(void** p, size_t* cap_bytes, size_t need) {
    if (*p && *cap_bytes >= need) return SGX_OK;
    if (*p) hipFree(*p);
    *p = nullptr;
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) {
        sgx_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        *cap_bytes = 0;
        return SGX_E_NOMEM;
    }
    *cap_bytes = need;
    return SGX_OK;
}

static int acquire_four_step(sgx_ctx* c, SgxSig x, size_t n_samples, const int32_t* prn0,
                             int32_t n_prn, int32_t n_blocks, int32_t noncoh, double* carrFreq, double* codePhase,
                             double* peakMetric, int32_t* freqBin, int32_t* fineIdx, bool* handled, bool defer = false);
static int acquire_passes(sgx_ctx* c, SgxSig x, size_t n_samples, const int32_t* prn0, int32_t n_prn, int32_t n_blocks,
                          int32_t noncoh, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin,
                          int32_t* fineIdx);
static int acquire_fine(sgx_ctx* c, SgxSig x, size_t n_samples, const std::vector<int>& det_prn,
                        const std::vector<int>& det_phase, const std::vector<int>& det_slot, long long* d_sum,
                        double* carrFreq, double* codePhase, int32_t* fineIdx);

static int acquire_any(sgx_ctx* c, SgxSig x, size_t n_samples, const int32_t* prn0, int32_t n_prn, int32_t n_blocks,
                       int32_t noncoh, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin,
                       int32_t* fineIdx) {
    // the four-step path (sub-transforms in registers and LDS, shifted forward spectra) where it applies
    bool handled = false;
    const int rc4 = acquire_four_step(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, carrFreq, codePhase, peakMetric,
                                      freqBin, fineIdx, &handled);
    if (handled) return rc4;
    return acquire_passes(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, carrFreq, codePhase, peakMetric, freqBin, fineIdx);
}

extern "C" int sgx_acquire(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n_samples, const int32_t* prn0,
                           int32_t n_prn, int32_t n_blocks, int32_t noncoh, double* carrFreq, double* codePhase,
                           double* peakMetric, int32_t* freqBin, int32_t* fineIdx) {
    SGX_CHECK_ARG(c && r && prn0 && carrFreq && codePhase && peakMetric && freqBin && fineIdx);
    SGX_CHECK_ARG(n_prn >= 1 && n_prn <= 32 && n_blocks >= 1 && n_blocks <= 64);
    for (int i = 0; i < n_prn; ++i) SGX_CHECK_ARG(prn0[i] >= 0 && prn0[i] < 32);
    const long long N = c->n_code;
    if (offset > r->n || n_samples > r->n - offset || (long long)n_samples < (long long)n_blocks * N) {
        sgx_set_error("record window too short: %zu samples at offset %zu, %lld needed for the coarse search",
                      n_samples, offset, (long long)n_blocks * N);
        return SGX_E_RANGE;
    }
    {
        const int rq = sgx_if_require(r, offset + n_samples);   // a record that is still streaming in
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    SgxSig x;
    x.i8 = r->d + offset;
    x.f64 = nullptr;
    return acquire_any(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, carrFreq, codePhase, peakMetric, freqBin, fineIdx);
}

// acquire() on a signal that is not int8 (acquisition.py:55-59 takes whatever real dtype numpy hands it): the caller's
// fp64 samples are copied to HBM and every kernel reads them instead of the int8 record; the arithmetic is the same
// fp64 arithmetic either way.
extern "C" int sgx_acquire_f64(sgx_ctx* c, const double* signal, size_t n_samples, const int32_t* prn0, int32_t n_prn,
                               int32_t n_blocks, int32_t noncoh, double* carrFreq, double* codePhase, double* peakMetric,
                               int32_t* freqBin, int32_t* fineIdx) {
    SGX_CHECK_ARG(c && signal && prn0 && carrFreq && codePhase && peakMetric && freqBin && fineIdx);
    SGX_CHECK_ARG(n_prn >= 1 && n_prn <= 32 && n_blocks >= 1 && n_blocks <= 64);
    for (int i = 0; i < n_prn; ++i) SGX_CHECK_ARG(prn0[i] >= 0 && prn0[i] < 32);
    const long long N = c->n_code;
    if ((long long)n_samples < (long long)n_blocks * N) {
        sgx_set_error("signal too short: %zu samples, %lld needed for the coarse search", n_samples, (long long)n_blocks * N);
        return SGX_E_RANGE;
    }
    SGX_HIP(hipSetDevice(c->device));
    const size_t need = sizeof(double) * (n_samples + 64);
    if (c->cap_sig64 < need) {
        if (c->d_sig64) hipFree(c->d_sig64);
        c->d_sig64 = nullptr;
        c->cap_sig64 = 0;
        if (hipMalloc((void**)&c->d_sig64, need) != hipSuccess) {
            sgx_set_error("hipMalloc of %zu signal bytes failed", need);
            return SGX_E_NOMEM;
        }
        c->cap_sig64 = need;
    }
    SGX_HIP(hipMemcpyAsync(c->d_sig64, signal, sizeof(double) * n_samples, hipMemcpyHostToDevice, c->stream));
    SGX_HIP(hipStreamSynchronize(c->stream));   // the caller may free `signal` on return
    SgxSig x;
    x.i8 = nullptr;
    x.f64 = c->d_sig64;
    return acquire_any(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, carrFreq, codePhase, peakMetric, freqBin, fineIdx);
}

// The round-1 path: one launch per radix pass, every Doppler bin mixed separately (any factorable samplesPerCode).
static int acquire_passes(sgx_ctx* c, SgxSig x, size_t n_samples, const int32_t* prn0, int32_t n_prn, int32_t n_blocks,
                          int32_t noncoh, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin,
                          int32_t* fineIdx) {
    const long long N = c->n_code;
    const sgx_settings& S = c->s;
    hipStream_t st = c->stream;

    // A4 frequency grid (acquisition.py:68,99-101)
    const int n_bins = (int)(nearbyint(S.acqSearchBand * 2) + 1);
    SGX_CHECK_ARG(n_bins >= 1 && n_bins <= ACQ_MAX_BINS);
    MixArgs ma;
    ma.n_bins = n_bins;
    ma.n_blocks = n_blocks;
    for (int k = 0; k < n_bins; ++k) ma.frq[k] = S.IF - S.acqSearchBand / 2 * 1000 + 500.0 * k;
    const double ts = 1.0 / S.samplingFreq;
    const double tc = 1.0 / S.codeFreqBasis;
    const int spc = (int)llround(S.samplingFreq / S.codeFreqBasis);   // acquisition.py:145

    int rc = sgx_fft_plan_create(&c->plan_code, N);
    if (rc != SGX_OK) return rc;

    // ---- scratch ------------------------------------------------------------------------------
    const int rows_fwd = n_blocks * n_bins;
    const int rows_per_prn = rows_fwd;
    SGX_CHECK_ARG(rows_per_prn <= ACQ_MAX_ROWS);
    int prn_chunk = ACQ_MAX_ROWS / rows_per_prn;
    if (prn_chunk < 1) prn_chunk = 1;
    if (prn_chunk > n_prn) prn_chunk = n_prn;
    const size_t row_bytes = sizeof(cplx) * (size_t)N;
    size_t work_rows = (size_t)prn_chunk * rows_per_prn;
    if (work_rows < (size_t)rows_fwd) work_rows = rows_fwd;
    if (work_rows < (size_t)n_prn) work_rows = n_prn;
    if ((rc = ensure_buf((void**)&c->d_work[0], &c->cap_w0, work_rows * row_bytes)) != SGX_OK) return rc;
    if ((rc = ensure_buf((void**)&c->d_work[1], &c->cap_w1, work_rows * row_bytes)) != SGX_OK) return rc;
    if ((rc = ensure_buf((void**)&c->d_fwd, &c->cap_fwd, (size_t)rows_fwd * row_bytes)) != SGX_OK) return rc;
    if ((rc = ensure_buf((void**)&c->d_codefd, &c->cap_code, (size_t)n_prn * row_bytes)) != SGX_OK) return rc;
    const size_t pow_need = noncoh ? work_rows * sizeof(double) * (size_t)N : (size_t)ACQ_MAX_ROWS * 64 * 12 + 4096;
    if ((rc = ensure_buf((void**)&c->d_pow, &c->cap_pow, pow_need)) != SGX_OK) return rc;

    char* dsm = (char*)c->d_small;
    char* hsm = (char*)c->h_small;
    // d_small layout: [0,8) sum | [64, 64+128) prn list | [256, ...) rowmax doubles | rowarg ints | second | fine
    long long* d_sum = (long long*)dsm;
    int* d_prn = (int*)(dsm + 64);
    double* d_rowmax = (double*)(dsm + 1024);
    int* d_rowarg = (int*)(dsm + 1024 + 8 * 4096);
    double* d_second = (double*)(dsm + 1024 + 12 * 4096);
    int* d_detprn = (int*)(dsm + 1024 + 12 * 4096 + 512);
    int* d_detph = d_detprn + 32;
    const int nblk_last = sgx_fft_last_pass_blocks(&c->plan_code);
    int2* d_map = (int2*)(dsm + 200000);
    SecondArgs* d_sa = (SecondArgs*)(dsm + 600000);

    // per-workgroup maxima of the fused last pass live in the (otherwise unused) power buffer
    double* d_pmax = c->d_pow;
    int* d_parg = (int*)(c->d_pow + (size_t)ACQ_MAX_ROWS * 64);
    SGX_CHECK_ARG(nblk_last <= 64);
    hipEventRecord(c->ev[0], st);
    SGX_HIP(hipMemsetAsync(d_sum, 0, 8, st));
    SGX_HIP(hipMemcpyAsync(d_prn, prn0, sizeof(int) * (size_t)n_prn, hipMemcpyHostToDevice, st));
    if (x.f64) acq_sum_f64_kernel<<<1, 1024, 0, st>>>(x.f64, (long long)n_samples, d_sum);
    else acq_sum_kernel<<<256, 256, 0, st>>>(x.i8, (long long)n_samples, d_sum);

    // ---- PRN-independent part: mix + forward FFTs ------------------------------------------------
    {
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows_fwd);
        acq_mix_kernel<<<grid, 256, 0, st>>>(x, c->d_work[0], N, ts, ma);
        cplx* res = nullptr;
        rc = sgx_fft_forward(&c->plan_code, c->d_work[0], c->d_work[1], rows_fwd, st, &res, N);
        if (rc != SGX_OK) return rc;
        SGX_HIP(hipMemcpyAsync(c->d_fwd, res, (size_t)rows_fwd * row_bytes, hipMemcpyDeviceToDevice, st));
    }
    // ---- code spectra ---------------------------------------------------------------------------
    {
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)n_prn);
        acq_code_kernel<<<grid, 256, 0, st>>>(c->d_codes, d_prn, c->d_work[0], N, ts, tc);
        cplx* res = nullptr;
        rc = sgx_fft_forward(&c->plan_code, c->d_work[0], c->d_work[1], n_prn, st, &res, N);
        if (rc != SGX_OK) return rc;
        SGX_HIP(hipMemcpyAsync(c->d_codefd, res, (size_t)n_prn * row_bytes, hipMemcpyDeviceToDevice, st));
    }

    // ---- correlation + peak search, PRN chunk by chunk ---------------------------------------------
    std::vector<int> det_prn, det_phase, det_slot;
    int status = SGX_OK;
    for (int i = 0; i < n_prn; ++i) {
        carrFreq[i] = 0.0;
        codePhase[i] = 0.0;
        peakMetric[i] = 0.0;
        freqBin[i] = -1;
        fineIdx[i] = -1;
    }
    const double inv_n = 1.0 / (double)N;
    for (int p0 = 0; p0 < n_prn && status == SGX_OK; p0 += prn_chunk) {
        const int np = (p0 + prn_chunk <= n_prn) ? prn_chunk : (n_prn - p0);
        const int rows = np * rows_per_prn;
        const int rows_out = noncoh ? np * n_bins : rows;
        cplx* res = nullptr;
        if (noncoh) {
            // extension path: the blocks' powers are summed per sample, so rows are materialised
            dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows);
            acq_mul_kernel<<<grid, 256, 0, st>>>(c->d_fwd, c->d_codefd, c->d_work[0], N, rows_per_prn, p0);
            rc = sgx_fft_forward(&c->plan_code, c->d_work[0], c->d_work[1], rows, st, &res, N);
            if (rc != SGX_OK) return rc;
            acq_power_kernel<<<rows_out, 256, 0, st>>>(res, c->d_pow, d_rowmax, d_rowarg, N, inv_n, n_bins, n_blocks, 1);
        } else {
            // reference path, fused: conj(X)*F formed in the first radix pass, |.|^2 and the per-workgroup
            // maxima taken in the last one; no product rows, no power rows
            FftFuse fu;
            fu.mul_x = c->d_fwd;
            fu.mul_f = c->d_codefd;
            fu.rows_per_prn = rows_per_prn;
            fu.prn_base = p0;
            fu.pmax = d_pmax;
            fu.parg = d_parg;
            fu.inv_n = inv_n;
            rc = sgx_fft_forward_fused(&c->plan_code, c->d_work[0], c->d_work[1], rows, st, &res, N, &fu);
            if (rc != SGX_OK) return rc;
            acq_rowmax_finish_kernel<<<rows, 64, 0, st>>>(d_pmax, d_parg, nblk_last, d_rowmax, d_rowarg);
        }
        double* h_rowmax = (double*)(hsm + 1024);
        int* h_rowarg = (int*)(hsm + 1024 + 8 * 4096);
        SGX_HIP(hipMemcpyAsync(h_rowmax, d_rowmax, sizeof(double) * (size_t)rows_out, hipMemcpyDeviceToHost, st));
        SGX_HIP(hipMemcpyAsync(h_rowarg, d_rowarg, sizeof(int) * (size_t)rows_out, hipMemcpyDeviceToHost, st));
        SGX_HIP(hipStreamSynchronize(st));

        // host: block choice (A7), global peak (A8), exclusion list (A8b)
        SecondArgs sa;
        double peak[32];
        int cph[32], fbi[32];
        for (int pi = 0; pi < 32; ++pi) sa.row[pi] = -1, sa.lo0[pi] = sa.hi0[pi] = sa.lo1[pi] = sa.hi1[pi] = 0;
        for (int pi = 0; pi < np; ++pi) {
            double gmax = -1.0;
            int gk = 0, gc = 0, grow = 0;
            bool have = false;
            for (int k = 0; k < n_bins; ++k) {
                int row;
                if (noncoh) {
                    row = pi * n_bins + k;
                } else {
                    int best = 0;   // acquisition.py:129-133 generalised left to right, later block wins ties
                    for (int b = 1; b < n_blocks; ++b) {
                        const double vb = h_rowmax[(pi * n_blocks + best) * n_bins + k];
                        const double vn = h_rowmax[(pi * n_blocks + b) * n_bins + k];
                        if (!(vb > vn)) best = b;
                    }
                    row = (pi * n_blocks + best) * n_bins + k;
                }
                const double v = h_rowmax[row];
                const int a = h_rowarg[row];
                if (!have || v > gmax) {
                    gmax = v;
                    gk = k;          // first row attaining the maximum (results.max(1).argmax())
                    gc = a;
                    grow = row;
                    have = true;
                } else if (v == gmax && a < gc) {
                    gc = a;          // results.max(0).argmax(): first column attaining the maximum
                }
            }
            peak[pi] = gmax;
            cph[pi] = gc;
            fbi[pi] = gk;
            const int e1 = gc - spc, e2 = gc + spc;
            sa.row[pi] = grow;
            if (e1 <= 0) {
                if ((long long)N + e1 + 1 > N) {   // index N would be read: the reference's IndexError (Q5)
                    sgx_set_error("IndexError: index %lld is out of bounds for axis 1 with size %lld "
                                  "(PRN index %d, codePhase %d; reference acquisition.py:152-162)",
                                  N, N, prn0[p0 + pi], gc);
                    status = SGX_E_INDEX;
                    sa.row[pi] = -1;
                    break;
                }
                sa.lo0[pi] = e2;
                sa.hi0[pi] = (int)(N + e1 + 1);
            } else if (e2 >= N - 1) {
                const int lo = (int)(e2 - N);
                if (lo < 0) {   // arange starts at -1: numpy wraps it to N-1
                    sa.lo0[pi] = 0;
                    sa.hi0[pi] = e1;
                    sa.lo1[pi] = (int)N - 1;
                    sa.hi1[pi] = (int)N;
                } else {
                    sa.lo0[pi] = lo;
                    sa.hi0[pi] = e1;
                }
            } else {
                sa.lo0[pi] = 0;
                sa.hi0[pi] = e1 + 1;
                sa.lo1[pi] = e2;
                sa.hi1[pi] = (int)N;
            }
        }
        if (status != SGX_OK) break;
        if (noncoh) {
            SGX_HIP(hipMemcpyAsync(d_sa, &sa, sizeof(sa), hipMemcpyHostToDevice, st));
            SGX_HIP(hipMemsetAsync(d_second, 0, sizeof(double) * 32, st));
            acq_second_kernel<<<dim3((unsigned)np, SEC_SPLIT), 256, 0, st>>>(c->d_pow, d_second, N, d_sa);
        } else {
            // recompute only the np rows the second-peak search reads (one per PRN)
            int2* h_map = (int2*)(hsm + 200000);
            for (int pi = 0; pi < np; ++pi) {
                h_map[pi] = make_int2(sa.row[pi] % rows_per_prn, p0 + pi);   // row = (pi*blocks + b)*bins + k
                sa.row[pi] = pi;
            }
            SGX_HIP(hipMemcpyAsync(d_map, h_map, sizeof(int2) * (size_t)np, hipMemcpyHostToDevice, st));
            FftFuse fu;
            fu.mul_x = c->d_fwd;
            fu.mul_f = c->d_codefd;
            fu.row_map = d_map;
            cplx* r2 = nullptr;
            rc = sgx_fft_forward_fused(&c->plan_code, c->d_work[0], c->d_work[1], np, st, &r2, N, &fu);
            if (rc != SGX_OK) return rc;
            SGX_HIP(hipMemcpyAsync(d_sa, &sa, sizeof(sa), hipMemcpyHostToDevice, st));
            SGX_HIP(hipMemsetAsync(d_second, 0, sizeof(double) * 32, st));
            acq_second_cplx_kernel<<<dim3((unsigned)np, SEC_SPLIT), 256, 0, st>>>(r2, d_second, N, inv_n, d_sa);
        }
        double* h_second = (double*)(hsm + 1024 + 12 * 4096);
        SGX_HIP(hipMemcpyAsync(h_second, d_second, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, st));
        SGX_HIP(hipStreamSynchronize(st));
        for (int pi = 0; pi < np; ++pi) {
            const int o = p0 + pi;
            const double ratio = peak[pi] / h_second[pi];
            peakMetric[o] = ratio;
            freqBin[o] = fbi[pi];
            if (ratio > S.acqThreshold) {
                det_prn.push_back(prn0[o]);
                det_phase.push_back(cph[pi]);
                det_slot.push_back(o);
            }
        }
    }
    hipEventRecord(c->ev[1], st);
    if (status != SGX_OK) {
        hipStreamSynchronize(st);
        return status;
    }

    // ---- fine frequency search (acquisition.py:167-193) -----------------------------------------------
    {
        const int rcf = acquire_fine(c, x, n_samples, det_prn, det_phase, det_slot, d_sum, carrFreq, codePhase, fineIdx);
        if (rcf != SGX_OK) return rcf;
    }
    hipEventElapsedTime(&c->timing.acq_coarse_ms, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&c->timing.acq_fine_ms, c->ev[1], c->ev[2]);
    hipEventElapsedTime(&c->timing.acquire_ms, c->ev[0], c->ev[2]);
    return SGX_OK;
}

// Fine frequency search (acquisition.py:167-193) for the detected PRNs; records event ev[2] and synchronises.
static int acquire_fine(sgx_ctx* c, SgxSig x, size_t n_samples, const std::vector<int>& det_prn,
                        const std::vector<int>& det_phase, const std::vector<int>& det_slot, long long* d_sum,
                        double* carrFreq, double* codePhase, int32_t* fineIdx) {
    hipStream_t st = c->stream;
    const sgx_settings& S = c->s;
    const long long N = c->n_code;
    const double ts = 1.0 / S.samplingFreq;
    char* dsm = (char*)c->d_small;
    char* hsm = (char*)c->h_small;
    int* d_detprn = (int*)(dsm + 1024 + 12 * 4096 + 512);
    int* d_detph = d_detprn + 32;
    double* d_pv = (double*)(dsm + 65536);
    long long* d_pi = nullptr;
    int rc = SGX_OK;
    const int n_det = (int)det_prn.size();
    if (n_det > 0) {
        const long long len = 10 * N;
        const long long npts = 8ll << (long long)ceil(log2((double)len));
        const long long uniq = (long long)ceil((double)(npts + 1) / 2.0);
        for (int d = 0; d < n_det; ++d) {
            if ((long long)det_phase[d] + len > (long long)n_samples) {
                sgx_set_error("fine search needs codePhase + 10 ms = %lld samples, record window has %zu "
                              "(reference acquisition.py:177 would fail to broadcast)",
                              (long long)det_phase[d] + len, n_samples);
                return SGX_E_RANGE;
            }
        }
        rc = sgx_fft_plan_create(&c->plan_fine, npts);
        if (rc != SGX_OK) return rc;
        const int n_rows = (n_det + 1) / 2;   // two real signals per complex row
        if ((rc = ensure_buf((void**)&c->d_fine[0], &c->cap_f0, (size_t)n_rows * sizeof(cplx) * (size_t)npts)) != SGX_OK)
            return rc;
        if ((rc = ensure_buf((void**)&c->d_fine[1], &c->cap_f1, (size_t)n_rows * sizeof(cplx) * (size_t)npts)) != SGX_OK)
            return rc;
        const double tc1 = 1.0 / S.codeFreqBasis;
        const char* fv1 = getenv("SGX_ACQ_FINE_V1");
        const bool fine2 = sgx_fft_fine_supported(npts) && !(fv1 && fv1[0] == '1');
        double mean = 0.0;
        if (!fine2) {
            long long h_sum = 0;
            SGX_HIP(hipMemcpyAsync(&h_sum, d_sum, 8, hipMemcpyDeviceToHost, st));
            SGX_HIP(hipStreamSynchronize(st));
            double h_sumd;
            memcpy(&h_sumd, &h_sum, 8);
            mean = (x.f64 ? h_sumd : (double)h_sum) / (double)n_samples;   // longSignal.mean(), acquisition.py:59
        }
        int nblk = 256;
        double* h_pv = (double*)(hsm + 65536);
        long long* h_pi = (long long*)(hsm + 400000);   // (behind the row-map area at 200000)
        d_pi = (long long*)(dsm + 400000);
        if (fine2) {
            // two kernels with LDS-resident sub-transforms, input built on the fly (the mean comes from the device-side
            // sum: no host look), arg-max fused (sgx_fft.hip)
            nblk = sgx_fft_fine_partials();
            rc = sgx_fft_fine_search(&c->plan_fine, x, c->d_codes, det_prn.data(), det_phase.data(), n_det, len, d_sum, (double)n_samples, ts,
                                     tc1, c->d_fine[0], 4, uniq - 5, d_pv, d_pi, st);
            if (rc != SGX_OK) return rc;
        } else {
            SGX_HIP(hipMemcpyAsync(d_detprn, det_prn.data(), sizeof(int) * (size_t)n_det, hipMemcpyHostToDevice, st));
            SGX_HIP(hipMemcpyAsync(d_detph, det_phase.data(), sizeof(int) * (size_t)n_det, hipMemcpyHostToDevice, st));
            dim3 grid((unsigned)((len + 255) / 256), (unsigned)n_rows);
            acq_fine_prep_kernel<<<grid, 256, 0, st>>>(x, c->d_codes, c->d_fine[0], len, npts, mean, ts, tc1, d_detprn,
                                                       d_detph, n_det);
            cplx* res = nullptr;
            rc = sgx_fft_forward(&c->plan_fine, c->d_fine[0], c->d_fine[1], n_rows, st, &res, len);
            if (rc != SGX_OK) return rc;
            dim3 g2((unsigned)nblk, (unsigned)n_det);
            acq_fine_argmax_kernel<<<g2, 256, 0, st>>>(res, npts, 4, uniq - 5, d_pv, d_pi);
        }
        SGX_HIP(hipMemcpyAsync(h_pv, d_pv, sizeof(double) * (size_t)n_det * nblk, hipMemcpyDeviceToHost, st));
        SGX_HIP(hipMemcpyAsync(h_pi, d_pi, sizeof(long long) * (size_t)n_det * nblk, hipMemcpyDeviceToHost, st));
        hipEventRecord(c->ev[2], st);
        SGX_HIP(hipStreamSynchronize(st));
        for (int d = 0; d < n_det; ++d) {
            double bv = -1.0;
            long long bi = 0;
            for (int b = 0; b < nblk; ++b) {
                const double v = h_pv[d * nblk + b];
                const long long i = h_pi[d * nblk + b];
                if (v > bv || (v == bv && i < bi)) {
                    bv = v;
                    bi = i;
                }
            }
            const long long m = bi - 4;   // index inside the [4:uniq-5] slice (acquisition.py:187)
            const int o = det_slot[d];
            carrFreq[o] = ((double)m * S.samplingFreq) / (double)npts;   // acquisition.py:189-191 (Q3)
            codePhase[o] = (double)det_phase[d];
            fineIdx[o] = (int)m;
        }
    } else {
        hipEventRecord(c->ev[2], st);
        SGX_HIP(hipStreamSynchronize(st));
    }
    return SGX_OK;
}

// Peak logic of one PRN (acquisition.py:129-162) in pieces a wave can share.
// One bin's candidate: block choice (A7) and its row's maximum / first index.
struct AcqCand {
    double v;
    int k, a, b;
};
// (the row maxima come from other workgroups of the SAME launch, acq_rowmax_peak_kernel: written and read past the
// per-XCD L2s with device-scope accesses, no cache write-back or invalidation)
#define ACQ_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ACQ_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ static inline AcqCand acq_peak_bin(const double* __restrict__ rowmax, const int* __restrict__ rowarg, int n_bins,
                                              int n_blocks, bool noncoh, int k) {
    int row = k, bsel = 0;
    if (!noncoh) {
        int best = 0;   // acquisition.py:129-133 generalised left to right, later block wins ties
        for (int b = 1; b < n_blocks; ++b) {
            const double vb = ACQ_LD(rowmax + best * n_bins + k);
            const double vn = ACQ_LD(rowmax + b * n_bins + k);
            if (!(vb > vn)) best = b;
        }
        row = best * n_bins + k;
        bsel = best;
    }
    AcqCand c;
    c.v = ACQ_LD(rowmax + row);
    c.a = ACQ_LD(rowarg + row);
    c.k = k;
    c.b = bsel;
    return c;
}
// A scan over the bins in ascending k keeps: the maximum, the FIRST bin attaining it (results.max(1).argmax()), that
// bin's block, and the SMALLEST column among the bins attaining it (results.max(0).argmax()) - A8.  The same as a
// combination of two partial scans (k < 0: an empty one):
__device__ static inline AcqCand acq_peak_join(const AcqCand& x, const AcqCand& y) {
    if (y.k < 0) return x;
    if (x.k < 0) return y;
    if (x.v > y.v) return x;
    if (y.v > x.v) return y;
    AcqCand c = (x.k < y.k) ? x : y;
    c.a = x.a < y.a ? x.a : y.a;
    return c;
}
// The exclusion list around the peak's code phase (A8b, acquisition.py:135-162).  Returns 1 where the reference raises
// IndexError (Q5), else 0.
__device__ static inline int acq_peak_ranges(int gc, long long N, int spc, int* lo0, int* hi0, int* lo1, int* hi1) {
    *lo0 = *hi0 = *lo1 = *hi1 = 0;
    const int e1 = gc - spc, e2 = gc + spc;
    if (e1 <= 0) {
        if ((long long)N + e1 + 1 > N) return 1;   // index N would be read: the reference's IndexError (Q5)
        *lo0 = e2;
        *hi0 = (int)(N + e1 + 1);
    } else if (e2 >= N - 1) {
        const int lo = (int)(e2 - N);
        if (lo < 0) {   // arange starts at -1: numpy wraps it to N-1
            *lo0 = 0;
            *hi0 = e1;
            *lo1 = (int)N - 1;
            *hi1 = (int)N;
        } else {
            *lo0 = lo;
            *hi0 = e1;
        }
    } else {
        *lo0 = 0;
        *hi0 = e1 + 1;
        *lo1 = e2;
        *hi1 = (int)N;
    }
    return 0;
}

// The same for every PRN of a call on the device: fills the second-peak search's arguments and row map, so the host
// looks at the coarse search once (after the second peaks).
struct PeakOut {
    double peak[32];
    int cph[32], fbi[32];
    int index_error[32];
};
// The coarse search's outcome, written by one small kernel straight into a coherent pinned page: the host spins on
// `seq` instead of sleeping in hipStreamSynchronize behind two device-to-host copies (~55 us -> ~10 us between the last
// coarse kernel and the first fine one).
struct CoarseLook {
    PeakOut po;
    double second[32];
    unsigned long long seq;
    // device-led fine search (round 4): the detections the publish kernel found (in PRN order, as the reference's loop
    // finds them), and what the fine search made of them - the host looks ONCE, at seq2
    int n_det;
    int range_error;          // a detection's fine window (code phase + 10 ms) leaves the record: 1 + its slot
    int det_slot[32];         // position in the call's PRN list
    int det_phase[32];
    long long fine_bi[32];    // arg-max of the 2^22-point magnitude spectrum over [4, uniq - 5)
    unsigned long long seq2;
};
static_assert(sizeof(CoarseLook) <= 4096, "one pinned page");

// det (device memory, read by the fine kernels): [0] n_det, [1 + d] PRN index, [33 + d] code phase
// SAME_LAUNCH: the peaks were written by other waves of this launch (device-scope stores): read them the same way.
template <bool SAME_LAUNCH>
__device__ __forceinline__ void acq_publish_body(const PeakOut* __restrict__ po, const double* __restrict__ second, int n_prn,
                                                 CoarseLook* __restrict__ host, const int* __restrict__ prn_list,
                                                 double threshold, long long fine_len, long long n_samples,
                                                 int* __restrict__ det, int t) {
    const int* src = reinterpret_cast<const int*>(po);
    int* dst = reinterpret_cast<int*>(&host->po);
    for (int i = t; i < (int)(sizeof(PeakOut) / sizeof(int)); i += 64) dst[i] = SAME_LAUNCH ? ACQ_LD(src + i) : src[i];
    double sec = 0.0, pk = 0.0;
    int ie = 0, cp = 0;
    if (t < n_prn) {
        sec = SAME_LAUNCH ? ACQ_LD(second + t) : second[t];
        pk = SAME_LAUNCH ? ACQ_LD(po->peak + t) : po->peak[t];
        ie = SAME_LAUNCH ? ACQ_LD(po->index_error + t) : po->index_error[t];
        cp = SAME_LAUNCH ? ACQ_LD(po->cph + t) : po->cph[t];
        host->second[t] = sec;
    }
    if (det) {
        // acquisition.py:164-166: detected iff peak / second peak > acqThreshold; the list in ascending PRN position
        const bool hit = t < n_prn && ie == 0 && (pk / sec) > threshold;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        const int d = __builtin_popcountll(m & ((1ull << t) - 1ull));
        const bool out = hit && (long long)cp + fine_len > n_samples;   // (the reference would fail to broadcast)
        const unsigned long long mo = __builtin_amdgcn_ballot_w64(out);
        if (hit) {
            det[1 + d] = prn_list[t];
            det[33 + d] = cp;
            host->det_slot[d] = t;
            host->det_phase[d] = cp;
        }
        if (t == 0) {
            det[80] = 0;                                    // fine_rows_kernel's arrival counter
            det[0] = mo ? 0 : __builtin_popcountll(m);      // (an error: the fine kernels have nothing to do)
            host->n_det = __builtin_popcountll(m);
            host->range_error = mo ? 1 + __builtin_ctzll(mo) : 0;
            host->seq = 0ull;   // device-led: `host` is a device-side copy of the page; fine_rows_kernel's last workgroup
        }                       // copies it to the real one and the host waits for seq2
    }
}

__global__ __launch_bounds__(64) void acq_publish_kernel(const PeakOut* __restrict__ po, const double* __restrict__ second,
                                                         int n_prn, CoarseLook* __restrict__ host, unsigned long long seq,
                                                         const int* __restrict__ prn_list, double threshold,
                                                         long long fine_len, long long n_samples, int* __restrict__ det) {
    const int t = threadIdx.x;
    acq_publish_body<false>(po, second, n_prn, host, prn_list, threshold, fine_len, n_samples, det, t);
    if (det) return;
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits for acq_publish_kernel's `seq`.  Spins (bounded), then falls back to the stream synchronisation, after which the
// page is complete in any case.
static int coarse_look_wait(sgx_ctx* c, unsigned long long seq, bool second = false) {
    const CoarseLook* h0 = (const CoarseLook*)c->h_look;
    const unsigned long long* word = second ? &h0->seq2 : &h0->seq;
    const char* sp = getenv("SGX_ACQ_SPIN");
    if (!(sp && sp[0] == '0')) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; ++it) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return SGX_OK;
            if ((it & 1023u) == 1023u &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
                break;
        }
    }
    SGX_HIP(hipStreamSynchronize(c->stream));
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
        sgx_set_error("acquisition: the search's result page was not written");
        return SGX_E_HIP;
    }
    return SGX_OK;
}

// One WAVE per PRN, a lane per Doppler bin (one lane per PRN scanning its rows was a chain of dependent loads: 12 us).
__device__ __forceinline__ void acq_peak_one(const double* __restrict__ rowmax, const int* __restrict__ rowarg, int pi,
                                             int lane, int n_prn, int out_per_prn, int n_bins, int n_blocks, int noncoh,
                                             long long N, int spc, PeakOut* __restrict__ po, SecondArgs* __restrict__ sa,
                                             int2* __restrict__ row_map) {
    if (pi >= n_prn) {
        if (lane == 0) {
            sa->row[pi] = -1;
            sa->lo0[pi] = sa->hi0[pi] = sa->lo1[pi] = sa->hi1[pi] = 0;
        }
        return;
    }
    const double* __restrict__ pm = rowmax + (long long)pi * out_per_prn;
    const int* __restrict__ pa = rowarg + (long long)pi * out_per_prn;
    AcqCand c;
    c.v = -1.0;
    c.k = -1;
    c.a = c.b = 0;
    for (int k = lane; k < n_bins; k += 64) c = acq_peak_join(c, acq_peak_bin(pm, pa, n_bins, n_blocks, noncoh != 0, k));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        AcqCand o;
        o.v = __shfl_down(c.v, off);
        o.k = __shfl_down(c.k, off);
        o.a = __shfl_down(c.a, off);
        o.b = __shfl_down(c.b, off);
        c = acq_peak_join(c, o);
    }
    if (lane != 0) return;
    int lo0, hi0, lo1, hi1;
    const int bad = acq_peak_ranges(c.a, N, spc, &lo0, &hi0, &lo1, &hi1);
    po->peak[pi] = c.v;
    po->cph[pi] = c.a;
    po->fbi[pi] = c.k;
    po->index_error[pi] = bad;
    sa->row[pi] = bad ? -1 : pi;
    sa->lo0[pi] = bad ? 0 : lo0;
    sa->hi0[pi] = bad ? 0 : hi0;
    sa->lo1[pi] = bad ? 0 : lo1;
    sa->hi1[pi] = bad ? 0 : hi1;
    if (noncoh) {
        for (int b = 0; b < n_blocks; ++b) row_map[pi * n_blocks + b] = make_int2(b * n_bins + c.k, pi);
    } else {
        row_map[pi] = make_int2(c.b * n_bins + c.k, pi);
    }
}

// acq_rowmax_finish_kernel and the peak step in one launch (round 4): a wave finishes one output row; the wave that
// finishes the LAST row of a PRN (arrival counter per PRN, zeroed by the call's set-up) goes on to that PRN's block
// choice, global peak and exclusion list.  The last PRN's wave also fills the unused slots of the second-peak arguments.
__global__ __launch_bounds__(64) void acq_rowmax_peak_kernel(const double* __restrict__ pmax, const int* __restrict__ parg,
                                                             int nblk, double* __restrict__ rowmax, int* __restrict__ rowarg,
                                                             int* __restrict__ arrived, int n_prn, int out_per_prn,
                                                             int n_bins, int n_blocks, int noncoh, long long N, int spc,
                                                             PeakOut* __restrict__ po, SecondArgs* __restrict__ sa,
                                                             int2* __restrict__ row_map) {
    const int row = blockIdx.x, lane = threadIdx.x;
    double best = -1.0;
    int arg = 0;
    for (int b = lane; b < nblk; b += 64) {
        const double v = pmax[(long long)row * nblk + b];
        const int i = parg[(long long)row * nblk + b];
        if (v > best || (v == best && i < arg)) {
            best = v;
            arg = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o);
        const int oi = __shfl_down(arg, o);
        if (ov > best || (ov == best && oi < arg)) {
            best = ov;
            arg = oi;
        }
    }
    const int pi = row / out_per_prn;
    int last = 0;
    if (lane == 0) {
        ACQ_ST(rowmax + row, best);
        ACQ_ST(rowarg + row, arg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both written through before the arrival is counted
        last = __hip_atomic_fetch_add(arrived + pi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == out_per_prn;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    acq_peak_one(rowmax, rowarg, pi, lane, n_prn, out_per_prn, n_bins, n_blocks, noncoh, N, spc, po, sa, row_map);
    if (pi == n_prn - 1 && n_prn + lane < 32)
        acq_peak_one(rowmax, rowarg, n_prn + lane, 0, n_prn, out_per_prn, n_bins, n_blocks, noncoh, N, spc, po, sa, row_map);
}

// The peak of one PRN on every lane of the wave (acq_peak_one's scan, result broadcast).
__device__ __forceinline__ AcqCand acq_peak_scan(const double* __restrict__ pm, const int* __restrict__ pa, int lane,
                                                 int n_bins, int n_blocks, int noncoh) {
    AcqCand c;
    c.v = -1.0;
    c.k = -1;
    c.a = c.b = 0;
    for (int k = lane; k < n_bins; k += 64) c = acq_peak_join(c, acq_peak_bin(pm, pa, n_bins, n_blocks, noncoh != 0, k));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        AcqCand o;
        o.v = __shfl_down(c.v, off);
        o.k = __shfl_down(c.k, off);
        o.a = __shfl_down(c.a, off);
        o.b = __shfl_down(c.b, off);
        c = acq_peak_join(c, o);
    }
    c.v = __shfl(c.v, 0);
    c.k = __shfl(c.k, 0);
    c.a = __shfl(c.a, 0);
    c.b = __shfl(c.b, 0);
    return c;
}

struct PublishArgs {
    CoarseLook* stage;          // device-side copy of the result page, or null: a publish kernel follows
    const int* prn_list;
    double threshold;
    long long fine_len, n_samples;
    int* det;
};

// Round 5: row maxima, peak, SECOND PEAK and the detection list in one launch, from the rows kernel's per-residue
// (maximum, maximum of the others, first index) triples - the winning row is not transformed again (two launches, 39 us
// of the 8-rank shard of config 4) and no publish kernel follows (10 us).  A wave finishes one output row; the wave that
// finishes the last row of a PRN goes on to that PRN's block choice, global peak, exclusion list (acquisition.py:129-162)
// and the maximum over the allowed indices: the excluded ones are fewer than `nres` consecutive indices, at most one per
// residue, so a residue contributes its maximum if that is allowed and else the maximum of its others - the same powers
// the first pass formed.  The wave that finishes the last PRN decides the detections (acquisition.py:164-166).
__global__ __launch_bounds__(64) void acq_rowtop2_peak_kernel(const double* __restrict__ b1, const double* __restrict__ b2,
                                                              const int* __restrict__ i1, int nres,
                                                              double* __restrict__ rowmax, int* __restrict__ rowarg,
                                                              int* __restrict__ arrived, int n_prn, int out_per_prn,
                                                              int n_bins, int n_blocks, int noncoh, long long N, int spc,
                                                              PeakOut* __restrict__ po, double* __restrict__ second,
                                                              PublishArgs pub) {
    const int row = blockIdx.x, lane = threadIdx.x;
    double best = -1.0;
    int arg = 0;
    for (int r = lane; r < nres; r += 64) {
        const double v = b1[(long long)row * nres + r];
        const int i = i1[(long long)row * nres + r];
        if (v > best || (v == best && i < arg)) {
            best = v;
            arg = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o);
        const int oi = __shfl_down(arg, o);
        if (ov > best || (ov == best && oi < arg)) {
            best = ov;
            arg = oi;
        }
    }
    const int pi = row / out_per_prn;
    int last = 0;
    if (lane == 0) {
        ACQ_ST(rowmax + row, best);
        ACQ_ST(rowarg + row, arg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both written through before the arrival is counted
        last = __hip_atomic_fetch_add(arrived + pi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == out_per_prn;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    const AcqCand c = acq_peak_scan(rowmax + (long long)pi * out_per_prn, rowarg + (long long)pi * out_per_prn, lane, n_bins,
                                    n_blocks, noncoh);
    int lo0, hi0, lo1, hi1;
    const int bad = acq_peak_ranges(c.a, N, spc, &lo0, &hi0, &lo1, &hi1);
    double sec = 0.0;
    if (!bad) {
        const long long wrow = ((long long)pi * out_per_prn + (noncoh ? c.k : c.b * n_bins + c.k)) * nres;
        for (int r = lane; r < nres; r += 64) {
            const int idx = i1[wrow + r];
            const double v1 = b1[wrow + r], v2 = b2[wrow + r];   // (both: three independent loads)
            const bool in = (idx >= lo0 && idx < hi0) || (idx >= lo1 && idx < hi1);
            sec = fmax(sec, in ? v1 : v2);
        }
    }
    for (int o = 32; o > 0; o >>= 1) sec = fmax(sec, __shfl_down(sec, o));
    int done = 0;
    if (lane == 0) {
        ACQ_ST(po->peak + pi, c.v);
        ACQ_ST(po->cph + pi, c.a);
        ACQ_ST(po->fbi + pi, c.k);
        ACQ_ST(po->index_error + pi, bad);
        ACQ_ST(second + pi, sec);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        done = __hip_atomic_fetch_add(arrived + 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == n_prn;
    }
    done = __builtin_amdgcn_readfirstlane(done);
    if (!done || !pub.stage) return;
    acq_publish_body<true>(po, second, n_prn, pub.stage, pub.prn_list, pub.threshold, pub.fine_len, pub.n_samples, pub.det,
                           lane);
}

// How the correlation batch of a call is cut (round 5).  Rows are ordered (PRN, block, bin) - coherent - or (PRN, bin, block)
// - non-coherent sums; a chunk is whole PRNs (prn_chunk of them) or, for non-coherent sums whose PRN does not fit half a
// chunk, ONE PRN's rows of a run of Doppler bins (bin_runs runs per PRN: a run is a batch of its own with fewer bins).
// The chunks alternate between `queues` HIP streams, each with its own intermediate of chunk_rows / queues rows: the columns
// kernel is bound by its stores and the rows kernel by its loads, and with two chunks in flight the one's stores overlap
// the other's loads (0.85 -> 0.77 ms for config 2, 3.21 -> 2.89 ms for config 4; both kernels move their bytes at 3-5 TB/s
// over the same fabric - the intermediate lives in the Infinity Cache - so a producer / consumer fusion has no more to win).
struct AcqPlan {
    int prn_chunk, bin_runs, bins_per_run, queues;
};
static AcqPlan acq_plan(int n_prn, int n_bins, int n_blocks, bool noncoh, int chunk_rows, int max_queues) {
    if (chunk_rows > ACQ_MAX_ROWS) chunk_rows = ACQ_MAX_ROWS;
    if (chunk_rows < 1) chunk_rows = 1;
    const int rows_per_prn = n_bins * n_blocks;
    AcqPlan p;
    p.bin_runs = 1;
    p.queues = (max_queues >= 2 && n_prn >= 2 && chunk_rows / 2 >= rows_per_prn) ? 2 : 1;
    if (max_queues >= 2 && p.queues == 1 && noncoh && rows_per_prn > chunk_rows / 2 && rows_per_prn <= chunk_rows && n_bins >= 2) {
        int runs = (rows_per_prn + chunk_rows / 2 - 1) / (chunk_rows / 2);
        if (runs > n_bins) runs = n_bins;
        if (runs >= 2) {
            p.bin_runs = runs;
            p.queues = 2;
        }
    }
    if (p.queues == 2 && p.bin_runs == 1) chunk_rows /= 2;
    p.prn_chunk = p.bin_runs > 1 ? 1 : chunk_rows / rows_per_prn;   // (a run of bins belongs to ONE PRN)
    if (p.prn_chunk < 1) p.prn_chunk = 1;
    if (p.prn_chunk > n_prn) p.prn_chunk = n_prn;
    if (p.queues == 2 && p.bin_runs == 1 && p.prn_chunk > (n_prn + 1) / 2) p.prn_chunk = (n_prn + 1) / 2;   // (both queues get work)
    p.bins_per_run = (n_bins + p.bin_runs - 1) / p.bin_runs;
    return p;
}
extern "C" int sgx_acquire_plan(int32_t n_prn, int32_t n_bins, int32_t n_blocks, int32_t noncoh, int32_t chunk_rows,
                                int32_t max_queues, int32_t* prn_chunk, int32_t* bin_runs, int32_t* bins_per_run,
                                int32_t* queues) {
    SGX_CHECK_ARG(n_prn >= 1 && n_bins >= 1 && n_blocks >= 1 && prn_chunk && bin_runs && bins_per_run && queues);
    const AcqPlan p = acq_plan(n_prn, n_bins, n_blocks, noncoh != 0, chunk_rows > 0 ? chunk_rows : ACQ_DEFAULT_CHUNK_ROWS, max_queues);
    *prn_chunk = p.prn_chunk;
    *bin_runs = p.bin_runs;
    *bins_per_run = p.bins_per_run;
    *queues = p.queues;
    return SGX_OK;
}

extern "C" int sgx_acquire_plan_limits(int32_t* default_chunk_rows, int32_t* max_rows) {
    SGX_CHECK_ARG(default_chunk_rows && max_rows);
    *default_chunk_rows = ACQ_DEFAULT_CHUNK_ROWS;
    *max_rows = ACQ_MAX_ROWS;
    return SGX_OK;
}

// The acquisition on the four-step transform (sgx_fft.hip): every 38192-point transform is two kernels with register-resident
// sub-transforms, the mixed-signal spectra are computed once per (block, phi) and read with a circular shift, results
// land where they are needed (no device-to-device copies) and the host looks at the device ONCE, at the very end of the
// call (round 4: peaks, second peaks, the detections and their fine-search results arrive in one pinned page), whatever the
// number of PRN chunks.
static int acquire_four_step(sgx_ctx* c, SgxSig x, size_t n_samples, const int32_t* prn0,
                             int32_t n_prn, int32_t n_blocks, int32_t noncoh, double* carrFreq, double* codePhase,
                             double* peakMetric, int32_t* freqBin, int32_t* fineIdx, bool* handled, bool defer) {
    *handled = false;
    const long long N = c->n_code;
    const sgx_settings& S = c->s;
    const char* v1 = getenv("SGX_ACQ_V1");
    if ((v1 && v1[0] == '1') || !sgx_fft4_supported(N)) return SGX_OK;
    const int n_bins = (int)(nearbyint(S.acqSearchBand * 2) + 1);
    if (n_bins < 1 || n_bins > ACQ_MAX_BINS) return SGX_OK;
    // f N / fs = shift + phi for every bin; the path needs few distinct phi
    PhiArgs pa;
    pa.n_phi = 0;
    std::vector<int2> bin_map((size_t)n_bins);
    for (int k = 0; k < n_bins; ++k) {
        const double f = S.IF - S.acqSearchBand / 2 * 1000 + 500.0 * k;   // A4 (acquisition.py:68,99-101)
        const double ratio = f * (double)N / S.samplingFreq;
        double sh = floor(ratio + 1e-9);
        double phi = ratio - sh;
        if (phi < 1e-9) phi = 0.0;
        int j = -1;
        for (int q = 0; q < pa.n_phi; ++q)
            if (fabs(pa.phi[q] - phi) < 1e-9) j = q;
        if (j < 0) {
            if (pa.n_phi == 4) return SGX_OK;   // too many distinct fractions: the direct path mixes every bin
            j = pa.n_phi;
            pa.phi[pa.n_phi++] = phi;
        }
        long long shm = (long long)sh % N;
        if (shm < 0) shm += N;
        bin_map[(size_t)k] = make_int2(j, (int)shm);
    }
    if (pa.n_phi >= n_bins && n_bins > 1) return SGX_OK;
    *handled = true;

    hipStream_t st = c->stream;
    const double ts = 1.0 / S.samplingFreq;
    const double tc = 1.0 / S.codeFreqBasis;
    const int spc = (int)llround(S.samplingFreq / S.codeFreqBasis);   // acquisition.py:145
    int rc = sgx_fft_plan_create(&c->plan_code, N);
    if (rc != SGX_OK) return rc;

    // ---- scratch ------------------------------------------------------------------------------
    const int n_phi = pa.n_phi;
    const int rows_fwd = n_blocks * n_phi;
    const int rows_per_prn = n_blocks * n_bins;
    SGX_CHECK_ARG(rows_per_prn <= ACQ_MAX_ROWS);
    // PRN chunks of ~350 rows: a chunk's intermediate (213 MB) then stays in the 256 MiB Infinity Cache between the
    // columns kernel that writes it and the rows kernel that reads it, and the next chunk overwrites it there.  With the
    // round-3 kernels - bound by their stores and by the dirty lines on their way out, not by instruction issue or LDS
    // any more - that is 0.94 -> 0.80 ms for config 2 and 3.43 -> 3.24 ms for config 4 (tools/acq_chunk_probe.py; the
    // round-2 kernels measured no difference).
    int chunk_rows = ACQ_DEFAULT_CHUNK_ROWS;
    {
        const char* ce = getenv("SGX_ACQ_CHUNK_ROWS");
        if (ce && atoi(ce) > 0) chunk_rows = atoi(ce);
    }
    const char* se = getenv("SGX_ACQ_STREAMS");
    const AcqPlan plan = acq_plan(n_prn, n_bins, n_blocks, noncoh != 0, chunk_rows, (se && se[0] == '1') ? 1 : 2);
    const bool two_q = plan.queues == 2;
    const int bin_runs = plan.bin_runs, prn_chunk = plan.prn_chunk;
    const size_t row_bytes = sizeof(cplx) * (size_t)N;
    size_t work_rows = (size_t)prn_chunk * rows_per_prn;
    if (work_rows < (size_t)(rows_fwd + n_prn)) work_rows = (size_t)(rows_fwd + n_prn);
    if (work_rows < (size_t)n_prn * (noncoh ? n_blocks : 1)) work_rows = (size_t)n_prn * (noncoh ? n_blocks : 1);
    if ((rc = ensure_buf((void**)&c->d_work[0], &c->cap_w0, work_rows * row_bytes)) != SGX_OK) return rc;
    if ((rc = ensure_buf((void**)&c->d_work[1], &c->cap_w1, work_rows * row_bytes)) != SGX_OK) return rc;
    if ((rc = ensure_buf((void**)&c->d_fwd, &c->cap_fwd, (size_t)(rows_fwd + n_prn) * row_bytes)) != SGX_OK) return rc;
    const int nblk = sgx_fft4_row_blocks();
    const int nres = sgx_fft4_residues();
    const int rows_out_all = n_prn * (noncoh ? n_bins : rows_per_prn);
    // Round 5: peak and second peak from ONE pass (acq_rowtop2_peak_kernel) when the exclusion list leaves out fewer than
    // `nres` consecutive indices (2 spc of them at most: any sampling rate below 111 MHz); SGX_ACQ_TOP2=0: the round-4
    // sequence, which transforms each PRN's winning row a second time
    const char* t2e = getenv("SGX_ACQ_TOP2");
    const bool top2 = 2 * spc + 1 <= nres && !(t2e && t2e[0] == '0');
    size_t pow_need = (size_t)rows_out_all * nblk * 12 + 4096;
    if (noncoh && pow_need < (size_t)n_prn * sizeof(double) * (size_t)N) pow_need = (size_t)n_prn * sizeof(double) * (size_t)N;
    // [per-workgroup maxima | their indices] or [per-residue maxima | second maxima | indices], [row maxima | row indices],
    // then (non-coherent, round-4 sequence) the second-peak power rows
    const size_t part_bytes = (((size_t)rows_out_all * (top2 ? (size_t)nres * 20 : (size_t)nblk * 12)) + 255) / 256 * 256;
    const size_t red_bytes = (part_bytes + (size_t)rows_out_all * 12 + 1023) / 256 * 256;
    if ((rc = ensure_buf((void**)&c->d_pow, &c->cap_pow, red_bytes + pow_need)) != SGX_OK) return rc;
    char* red = (char*)c->d_pow;
    double* d_pmax = (double*)red;
    int* d_parg = (int*)(red + (size_t)rows_out_all * nblk * 8);
    double* d_t2b1 = (double*)red;
    double* d_t2b2 = d_t2b1 + (size_t)rows_out_all * nres;
    int* d_t2i1 = (int*)(d_t2b2 + (size_t)rows_out_all * nres);
    double* d_rowmax = (double*)(red + part_bytes);
    int* d_rowarg = (int*)(red + part_bytes + (size_t)rows_out_all * 8);
    double* d_power = (double*)(red + red_bytes);

    char* dsm = (char*)c->d_small;
    char* hsm = (char*)c->h_small;
    long long* d_sum = (long long*)dsm;
    int* d_prn = (int*)(dsm + 64);
    double* d_second = (double*)(dsm + 1024 + 12 * 4096);
    int2* d_binmap = (int2*)(dsm + 1024);              // [n_bins <= 128]
    int2* d_map = (int2*)(dsm + 200000);
    int* d_arrived = (int*)(dsm + 51200);              // [32] rows finished per PRN (acq_rowmax_peak_kernel)

    hipEventRecord(c->ev[0], st);
    cplx* const d_codefd = c->d_fwd + (size_t)rows_fwd * (size_t)N;
    {
        // ---- set-up, record sum, mixed rows (n_blocks x n_phi, PRN independent) and code rows (n_prn): one launch for
        //      int8 records; then the forward spectra of all of them as ONE batch, straight into d_fwd = [forward | code]
        AcqSetup su;
        memset(&su, 0, sizeof(su));
        su.n_prn = n_prn;
        su.n_bins = n_bins;
        for (int i = 0; i < n_prn; ++i) su.prn[i] = prn0[i];
        for (int k = 0; k < n_bins; ++k) su.bin[k] = bin_map[(size_t)k];
        const char* fr0 = getenv("SGX_ACQ_FRONT");
        static_assert(ACQ_MAX_BINS <= 256, "the set-up workgroup has 256 threads");
        if (!x.f64 && !(fr0 && fr0[0] == '0')) {
            const int ph = c->acq_sum_phase & 1;
            long long* sum_now = (long long*)(dsm + 16) + ph;
            long long* sum_next = (long long*)(dsm + 16) + (ph ^ 1);
            if (!c->acq_sum_clean[ph]) SGX_HIP(hipMemsetAsync(sum_now, 0, 8, st));
            const unsigned gx = (unsigned)((N + 255) / 256);
            acq_front_kernel<<<(unsigned)(rows_fwd + n_prn) * gx + ACQ_SUM_WGS + 1, 256, 0, st>>>(
                su, x, pa, c->d_codes, c->d_work[1], N, rows_fwd, ts, tc, (long long)n_samples, d_prn, d_binmap, sum_now,
                sum_next, d_second, d_arrived);
            c->acq_sum_clean[ph] = false;
            c->acq_sum_clean[ph ^ 1] = true;
            c->acq_sum_phase = ph ^ 1;
            d_sum = sum_now;
        } else {
            acq_setup_kernel<<<1, 128, 0, st>>>(su, d_prn, d_binmap, d_sum, d_second, d_arrived);
            if (x.f64) acq_sum_f64_kernel<<<1, 1024, 0, st>>>(x.f64, (long long)n_samples, d_sum);
            else acq_sum_kernel<<<64, 256, 0, st>>>(x.i8, (long long)n_samples, d_sum);
            dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows_fwd);
            acq_mixphi_kernel<<<grid, 256, 0, st>>>(x, c->d_work[1], N, pa);
            dim3 grid2((unsigned)((N + 255) / 256), (unsigned)n_prn);
            acq_code_kernel<<<grid2, 256, 0, st>>>(c->d_codes, d_prn, c->d_work[1] + (size_t)rows_fwd * (size_t)N, N, ts, tc);
        }
        rc = sgx_fft4_forward(&c->plan_code, c->d_work[1], c->d_work[0], c->d_fwd, rows_fwd + n_prn, st, nullptr);
        if (rc != SGX_OK) return rc;
    }
    for (int i = 0; i < n_prn; ++i) {
        carrFreq[i] = 0.0;
        codePhase[i] = 0.0;
        peakMetric[i] = 0.0;
        freqBin[i] = -1;
        fineIdx[i] = -1;
    }
    // ---- correlation, all PRN chunks queued back to back; row maxima of every PRN collected on the device -----------
    const double inv_n = 1.0 / (double)N;
    const int out_per_prn = noncoh ? n_bins : rows_per_prn;
    hipStream_t st2 = st;
    if (two_q) {
        if (!c->acq_stream2) {
            // (into locals; the context gets them only when ALL exist - a half-made second queue would fail every later call)
            int least = 0, greatest = 0;
            SGX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            hipStream_t ns = nullptr;
            hipEvent_t ne[2] = {nullptr, nullptr};
            hipError_t ce = (c->priority == 0) ? hipStreamCreateWithFlags(&ns, hipStreamNonBlocking)
                                               : hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, c->priority < 0 ? greatest : least);
            for (int i = 0; i < 2 && ce == hipSuccess; ++i) ce = hipEventCreateWithFlags(&ne[i], hipEventDisableTiming);
            if (ce != hipSuccess) {
                for (int i = 0; i < 2; ++i)
                    if (ne[i]) hipEventDestroy(ne[i]);
                if (ns) hipStreamDestroy(ns);
                sgx_set_error("acquisition: the second queue could not be created: %s", hipGetErrorString(ce));
                return SGX_E_HIP;
            }
            c->acq_stream2 = ns;
            c->acq_ev2[0] = ne[0];
            c->acq_ev2[1] = ne[1];
        }
        st2 = c->acq_stream2;
        // (the second queue's intermediate is the buffer the forward transforms read: they are queued in front)
        SGX_HIP(hipEventRecord(c->acq_ev2[0], st));
        SGX_HIP(hipStreamWaitEvent(st2, c->acq_ev2[0], 0));
    }
    int chunk_no = 0;
    const int bins_per_run = plan.bins_per_run;
    for (int p0 = 0; p0 < n_prn; p0 += prn_chunk)
        for (int bin0 = 0; bin0 < n_bins; bin0 += bins_per_run, ++chunk_no) {
            const int np = (p0 + prn_chunk <= n_prn) ? prn_chunk : (n_prn - p0);
            const int nb = bin_runs == 1 ? n_bins : (bin0 + bins_per_run <= n_bins ? bins_per_run : n_bins - bin0);
            Fft4Fuse fu;
            fu.mul_x = c->d_fwd;
            fu.mul_f = d_codefd;
            fu.bin_map = d_binmap + bin0;    // (a run of bins is a batch of its own with fewer bins)
            fu.n_bins = nb;
            fu.n_phi = n_phi;
            fu.rows_per_prn = bin_runs == 1 ? rows_per_prn : nb * n_blocks;
            fu.prn_base = p0;
            fu.n_blocks = n_blocks;
            fu.blocks_fast = noncoh ? 1 : 0;
            const size_t out0 = ((size_t)p0 * out_per_prn + (size_t)bin0);
            if (top2) {
                fu.t2_b1 = d_t2b1 + out0 * nres;
                fu.t2_b2 = d_t2b2 + out0 * nres;
                fu.t2_i1 = d_t2i1 + out0 * nres;
            } else {
                fu.pmax = d_pmax + out0 * nblk;
                fu.parg = d_parg + out0 * nblk;
            }
            fu.inv_n = inv_n;
            fu.sum_blocks = noncoh ? n_blocks : 1;
            const int q = two_q ? (chunk_no & 1) : 0;
            rc = sgx_fft4_forward(&c->plan_code, nullptr, c->d_work[q], nullptr, (int64_t)np * fu.rows_per_prn, q ? st2 : st, &fu);
            if (rc != SGX_OK) {
                // (the second queue may still hold chunks that write d_work[1]: nothing of the next call may overtake them)
                if (two_q) hipStreamSynchronize(st2);
                return rc;
            }
        }
    if (two_q) {
        SGX_HIP(hipEventRecord(c->acq_ev2[1], st2));
        SGX_HIP(hipStreamWaitEvent(st, c->acq_ev2[1], 0));
    }
    // ---- the fine search is queued right behind the coarse one: the detections are decided on the device
    //      (acquisition.py:164-166) and the fine kernels read their list, so the host looks ONCE, at the very end ----------
    const unsigned long long seq = ++c->look_seq;
    const long long fine_len = 10 * N;
    const long long npts = 8ll << (long long)ceil(log2((double)fine_len));
    const long long uniq = (long long)ceil((double)(npts + 1) / 2.0);
    const char* fv1 = getenv("SGX_ACQ_FINE_V1");
    const char* dl0 = getenv("SGX_ACQ_DEVICE_LED");
    const bool device_led = sgx_fft_fine_supported(npts) && !(fv1 && fv1[0] == '1') && !(dl0 && dl0[0] == '0') && n_prn <= 32;
    int* d_det = (int*)(dsm + 640000);
    double* d_pv = (double*)(dsm + 65536);
    long long* d_pi = (long long*)(dsm + 400000);
    if (device_led) {
        // (before the last coarse kernels are queued: nothing of the host's between them and the fine kernels)
        rc = sgx_fft_plan_create(&c->plan_fine, npts);
        if (rc != SGX_OK) return rc;
        const int max_rows = (n_prn + 1) / 2;   // two real signals per complex row; only the detections' rows are touched
        if ((rc = ensure_buf((void**)&c->d_fine[0], &c->cap_f0, (size_t)max_rows * sizeof(cplx) * (size_t)npts)) != SGX_OK) return rc;
    }
    // (device-led: into a device-side copy of the page - a kernel that writes host memory ends with a flush the next one
    // waits for, 5 us in front of the fine search)
    CoarseLook* const d_stage = (CoarseLook*)(dsm + 700000);
    // ---- device: row maxima, then per PRN block choice, global peak, exclusion list, second peak -----------------------
    PeakOut* d_po = (PeakOut*)(dsm + 620000);
    SecondArgs* d_sa = (SecondArgs*)(dsm + 600000);
    if (top2) {
        PublishArgs pub;
        pub.stage = device_led ? d_stage : nullptr;
        pub.prn_list = d_prn;
        pub.threshold = S.acqThreshold;
        pub.fine_len = fine_len;
        pub.n_samples = (long long)n_samples;
        pub.det = d_det;
        acq_rowtop2_peak_kernel<<<rows_out_all, 64, 0, st>>>(d_t2b1, d_t2b2, d_t2i1, nres, d_rowmax, d_rowarg, d_arrived, n_prn,
                                                             out_per_prn, n_bins, n_blocks, noncoh, N, spc, d_po, d_second, pub);
    } else {
        acq_rowmax_peak_kernel<<<rows_out_all, 64, 0, st>>>(d_pmax, d_parg, nblk, d_rowmax, d_rowarg, d_arrived, n_prn,
                                                            out_per_prn, n_bins, n_blocks, noncoh, N, spc, d_po, d_sa, d_map);
        // the rows the second-peak search reads, transformed again
        const int rows2 = n_prn * (noncoh ? n_blocks : 1);
        Fft4Fuse fu;
        fu.mul_x = c->d_fwd;
        fu.mul_f = d_codefd;
        fu.bin_map = d_binmap;
        fu.row_map = d_map;
        fu.n_bins = n_bins;
        fu.n_phi = n_phi;
        fu.n_blocks = n_blocks;
        // the rows kernel folds each row's maximum over the exclusion list into d_second itself (the same powers, formed
        // by the same arithmetic, as the first pass: peak / second peak is a ratio of consistently rounded values);
        // neither the rows nor their powers are stored
        static_assert(sizeof(SecondArgs) == 5 * 32 * sizeof(int), "row / lo0 / hi0 / lo1 / hi1, 32 each");
        fu.sec = reinterpret_cast<const int*>(d_sa);
        fu.second_out = d_second;
        fu.inv_n = inv_n;
        fu.sum_blocks = noncoh ? n_blocks : 1;
        rc = sgx_fft4_forward(&c->plan_code, nullptr, c->d_work[0], nullptr, rows2, st, &fu);
        if (rc != SGX_OK) return rc;
    }
    if (!top2 || !device_led)
        acq_publish_kernel<<<1, 64, 0, st>>>(d_po, d_second, n_prn, device_led ? d_stage : (CoarseLook*)c->d_look, seq, d_prn,
                                             S.acqThreshold, fine_len, (long long)n_samples, device_led ? d_det : nullptr);
    // (an event between the coarse and the fine kernels holds the fine search back by 6-8 us: recorded on request only)
    const char* sev = getenv("SGX_ACQ_SPLIT_EVENT");   // (read per call, like every other SGX_ACQ_* knob)
    const bool split_event = sev && sev[0] == '1';
    if (split_event || !device_led) hipEventRecord(c->ev[1], st);
    SGX_HIP(hipGetLastError());
    if (device_led) {
        rc = sgx_fft_fine_search(&c->plan_fine, x, c->d_codes, nullptr, nullptr, n_prn, fine_len, d_sum, (double)n_samples, ts,
                                 1.0 / S.codeFreqBasis, c->d_fine[0], 4, uniq - 5, d_pv, d_pi, st, d_det,
                                 ((CoarseLook*)c->d_look)->fine_bi, &((CoarseLook*)c->d_look)->seq2, seq,
                                 reinterpret_cast<const int*>(d_stage), reinterpret_cast<int*>(c->d_look),
                                 (int)(offsetof(CoarseLook, fine_bi) / sizeof(int)));
        if (rc != SGX_OK) return rc;
        hipEventRecord(c->ev[2], st);
        SGX_HIP(hipGetLastError());
        // everything is queued; what the look needs to be decoded later (sgx_acquire_finish)
        AcqPending& P = c->acq_pending;
        P.mode = 1;
        P.seq = seq;
        P.n_prn = n_prn;
        for (int i = 0; i < n_prn; ++i) P.prn0[i] = prn0[i];
        P.N = N;
        P.npts = npts;
        P.fine_len = fine_len;
        P.n_samples = n_samples;
        if (defer) return SGX_OK;
        return sgx_acquire_finish(c, carrFreq, codePhase, peakMetric, freqBin, fineIdx);
    }
    rc = coarse_look_wait(c, seq);
    if (rc != SGX_OK) return rc;
    const CoarseLook* look = (const CoarseLook*)c->h_look;
    const PeakOut* h_po = &look->po;
    const double* h_second = look->second;
    const double* peak = h_po->peak;
    const int* cph = h_po->cph;
    const int* fbi = h_po->fbi;
    for (int pi = 0; pi < n_prn; ++pi) {
        if (h_po->index_error[pi]) {
            sgx_set_error("IndexError: index %lld is out of bounds for axis 1 with size %lld "
                          "(PRN index %d, codePhase %d; reference acquisition.py:152-162)",
                          N, N, prn0[pi], cph[pi]);
            return SGX_E_INDEX;
        }
    }
    std::vector<int> det_prn, det_phase, det_slot;
    for (int pi = 0; pi < n_prn; ++pi) {
        const double ratio = peak[pi] / h_second[pi];
        peakMetric[pi] = ratio;
        freqBin[pi] = fbi[pi];
        if (ratio > S.acqThreshold) {
            det_prn.push_back(prn0[pi]);
            det_phase.push_back(cph[pi]);
            det_slot.push_back(pi);
        }
    }
    rc = acquire_fine(c, x, n_samples, det_prn, det_phase, det_slot, d_sum, carrFreq, codePhase, fineIdx);
    if (rc != SGX_OK) return rc;
    hipEventElapsedTime(&c->timing.acquire_ms, c->ev[0], c->ev[2]);
    hipEventElapsedTime(&c->timing.acq_coarse_ms, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&c->timing.acq_fine_ms, c->ev[1], c->ev[2]);
    return SGX_OK;
}

// The host's ONE look at a device-led acquisition (queued by acquire_four_step; c->acq_pending says what was asked): waits
// for the result page's second word, then decodes peaks, detections and fine frequencies exactly as the eager call did.
int sgx_acquire_finish(sgx_ctx* c, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin, int32_t* fineIdx) {
    AcqPending& P = c->acq_pending;
    if (P.mode == 2) {   // (the search could not be deferred and ran eagerly: its outputs were kept)
        P.mode = 0;
        for (int i = 0; i < P.n_prn; ++i) {
            carrFreq[i] = P.res_carr[i];
            codePhase[i] = P.res_cph[i];
            peakMetric[i] = P.res_met[i];
            freqBin[i] = P.res_fb[i];
            fineIdx[i] = P.res_fi[i];
        }
        return P.rc;
    }
    if (P.mode != 1) {
        sgx_set_error("sgx_acquire_end: no acquisition is pending on this context");
        return SGX_E_ARG;
    }
    P.mode = 0;
    const sgx_settings& S = c->s;
    const int n_prn = P.n_prn;
    const long long N = P.N, npts = P.npts, fine_len = P.fine_len;
    const size_t n_samples = P.n_samples;
    const int* prn0 = P.prn0;
    for (int i = 0; i < n_prn; ++i) {
        carrFreq[i] = 0.0;
        codePhase[i] = 0.0;
        peakMetric[i] = 0.0;
        freqBin[i] = -1;
        fineIdx[i] = -1;
    }
    int rc = coarse_look_wait(c, P.seq, true);
    if (rc != SGX_OK) return rc;
    const CoarseLook* look = (const CoarseLook*)c->h_look;
    const PeakOut* h_po = &look->po;
    const double* h_second = look->second;
    const double* peak = h_po->peak;
    const int* cph = h_po->cph;
    const int* fbi = h_po->fbi;
    for (int pi = 0; pi < n_prn; ++pi) {
        if (h_po->index_error[pi]) {
            sgx_set_error("IndexError: index %lld is out of bounds for axis 1 with size %lld "
                          "(PRN index %d, codePhase %d; reference acquisition.py:152-162)",
                          N, N, prn0[pi], cph[pi]);
            return SGX_E_INDEX;
        }
    }
    int n_det_host = 0;
    for (int pi = 0; pi < n_prn; ++pi) {
        const double ratio = peak[pi] / h_second[pi];
        peakMetric[pi] = ratio;
        freqBin[pi] = fbi[pi];
        if (ratio > S.acqThreshold) ++n_det_host;
    }
    if (look->range_error) {
        const int o = look->range_error - 1;
        sgx_set_error("fine search needs codePhase + 10 ms = %lld samples, record window has %zu "
                      "(reference acquisition.py:177 would fail to broadcast)", (long long)cph[o] + fine_len, n_samples);
        return SGX_E_RANGE;
    }
    if (look->n_det != n_det_host) {   // (the same comparison on the same doubles: cannot differ)
        sgx_set_error("acquisition: device found %d detections, host %d", look->n_det, n_det_host);
        return SGX_E_HIP;
    }
    for (int d = 0; d < look->n_det; ++d) {
        const long long m = look->fine_bi[d] - 4;   // index inside the [4:uniq-5] slice (acquisition.py:187)
        const int o = look->det_slot[d];
        carrFreq[o] = ((double)m * S.samplingFreq) / (double)npts;   // acquisition.py:189-191 (Q3)
        codePhase[o] = (double)look->det_phase[d];
        fineIdx[o] = (int)m;
    }
    // (the result word is stored a moment before the last kernel retires: the device times below need its event)
    SGX_HIP(hipEventSynchronize(c->ev[2]));
    hipEventElapsedTime(&c->timing.acquire_ms, c->ev[0], c->ev[2]);
    // (SGX_ACQ_SPLIT_EVENT=1 records an event between the coarse and the fine kernels, which holds the fine search back by
    // 6-8 us; without it the split is NOT measured: NaN, not total / 0)
    const char* sev = getenv("SGX_ACQ_SPLIT_EVENT");
    if (sev && sev[0] == '1') {
        hipEventElapsedTime(&c->timing.acq_coarse_ms, c->ev[0], c->ev[1]);
        hipEventElapsedTime(&c->timing.acq_fine_ms, c->ev[1], c->ev[2]);
    } else {
        c->timing.acq_coarse_ms = __builtin_nanf("");
        c->timing.acq_fine_ms = __builtin_nanf("");
    }
    return SGX_OK;
}


// ================================ round 6: deferred acquisition, preRun on the device ================================
// The reference's caller (initialize.py:484-506) runs acquire -> preRun -> track and looks at each result in between.  A
// caller that only wants the tracking results can queue all three: sgx_acquire_begin queues the search and returns,
// sgx_track_chained (sgx_trk.hip) queues preRun - the kernel below - and the tracking kernel behind it and waits ONCE;
// sgx_acquire_end then decodes the search's page (no waiting left).  Outputs are those of the eager calls, bit for bit:
// the same kernels in the same order, and the kernel below repeats the host's arithmetic (one IEEE multiplication and
// division for carrFreq, one division for peakMetric, a stable descending sort).
// acquisition.py:259-306 on the device.  One wave; lane p = PRN index p of the 32-entry result arrays.
__global__ __launch_bounds__(64) void acq_prerun_kernel(const CoarseLook* __restrict__ stage, const long long* __restrict__ fine_bi,
                                                        const int* __restrict__ prn_list, int n_prn, double fs, double npts,
                                                        TrkChan* __restrict__ d_ch, int n_ch, long long skip_bytes,
                                                        long long rec_file_offset, int sample_bytes,
                                                        StepLook* __restrict__ look) {
    __shared__ double s_met[32], s_carr[32], s_cph[32];
    __shared__ int s_err;
    const int t = threadIdx.x;
    if (t < 32) {
        s_met[t] = 0.0;
        s_carr[t] = 0.0;
        s_cph[t] = 0.0;
    }
    if (t == 0) s_err = 0;
    __syncthreads();
    if (t < n_prn) {
        s_met[prn_list[t]] = stage->po.peak[t] / stage->second[t];   // acquisition.py:164
        if (stage->po.index_error[t]) atomicOr(&s_err, 2);
    }
    if (t == 0 && stage->range_error) atomicOr(&s_err, 2);
    __syncthreads();
    const int n_det = stage->n_det;
    if (t < n_det && s_err == 0) {
        const long long m = fine_bi[t] - 4;                          // acquisition.py:187-191 (Q3)
        const int p = prn_list[stage->det_slot[t]];
        s_carr[p] = ((double)m * fs) / npts;
        s_cph[p] = (double)stage->det_phase[t];
    }
    __syncthreads();
    // sorted(enumerate(peakMetric), key = metric, reverse = True): stable, descending (acquisition.py:289-290)
    int rank = 0, nan = 0;
    if (t < 32) {
        const double mine = s_met[t];
        nan = (mine != mine) ? 1 : 0;
        for (int q = 0; q < 32; ++q) {
            const double o = s_met[q];
            rank += (o > mine || (o == mine && q < t)) ? 1 : 0;
        }
    }
    const unsigned long long any_nan = __builtin_amdgcn_ballot_w64(nan != 0);
    const int count = __builtin_popcountll(__builtin_amdgcn_ballot_w64(t < 32 && s_carr[t] > 0.0));   // sum(carrFreq > 0)
    int flags = s_err | (any_nan ? 1 : 0);
    const int n_act = flags ? 0 : (count < n_ch ? count : n_ch);
    // channels that are off (acquisition.py:281-284); the lanes holding ranks < n_act then fill theirs
    if (t < n_ch) {
        d_ch[t].acquiredFreq = 0.0;
        d_ch[t].pos0 = 0;
        d_ch[t].prn = 0;
        d_ch[t].pad = 0;
        if (t < 32) {
            look->prn[t] = 0;
            look->acquiredFreq[t] = 0.0;
            look->codePhase[t] = 0.0;
        }
    }
    __syncthreads();
    int before = 0;
    if (t < 32 && rank < n_act) {
        const long long p0 = skip_bytes + (long long)s_cph[t] - rec_file_offset;   // tracking.py:107
        if (p0 < 0) before = 1;
        d_ch[rank].acquiredFreq = s_carr[t];
        d_ch[rank].pos0 = p0 / sample_bytes;
        d_ch[rank].prn = t + 1;
        d_ch[rank].pad = (int)(p0 % sample_bytes);
        look->prn[rank] = t + 1;
        look->acquiredFreq[rank] = s_carr[t];
        look->codePhase[rank] = s_cph[t];
    }
    if (__builtin_amdgcn_ballot_w64(before != 0)) {
        flags |= 4;
        __syncthreads();
        if (t < n_ch) d_ch[t].prn = 0;      // (nothing is tracked; the host reports the channel)
    }
    if (t == 0) {
        look->n_ch = n_ch;
        look->n_active = n_act;
        look->flags = flags;
    }
}

int sgx_prerun_enqueue(sgx_ctx* c, TrkChan* d_ch, int n_ch, long long skip_bytes, long long rec_file_offset, int sample_bytes) {
    const AcqPending& P = c->acq_pending;
    if (P.mode != 1 || n_ch < 1 || n_ch > 32) return SGX_E_DEFER;
    char* dsm = (char*)c->d_small;
    const CoarseLook* d_stage = (const CoarseLook*)(dsm + 700000);
    const int* d_prn = (const int*)(dsm + 64);
    StepLook* look = (StepLook*)((char*)c->d_look + SGX_STEP_LOOK_OFFSET);
    acq_prerun_kernel<<<1, 64, 0, c->stream>>>(d_stage, ((const CoarseLook*)c->d_look)->fine_bi, d_prn, P.n_prn,
                                               c->s.samplingFreq, (double)P.npts, d_ch, n_ch, skip_bytes, rec_file_offset,
                                               sample_bytes, look);
    SGX_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_acquire_begin(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n_samples, const int32_t* prn0,
                                 int32_t n_prn, int32_t n_blocks, int32_t noncoh) {
    SGX_CHECK_ARG(c && r && prn0);
    SGX_CHECK_ARG(n_prn >= 1 && n_prn <= 32 && n_blocks >= 1 && n_blocks <= 64);
    for (int i = 0; i < n_prn; ++i) SGX_CHECK_ARG(prn0[i] >= 0 && prn0[i] < 32);
    const long long N = c->n_code;
    if (offset > r->n || n_samples > r->n - offset || (long long)n_samples < (long long)n_blocks * N) {
        sgx_set_error("record window too short: %zu samples at offset %zu, %lld needed for the coarse search",
                      n_samples, offset, (long long)n_blocks * N);
        return SGX_E_RANGE;
    }
    {
        const int rq = sgx_if_require(r, offset + n_samples);
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    SgxSig x;
    x.i8 = r->d + offset;
    x.f64 = nullptr;
    AcqPending& P = c->acq_pending;
    P.mode = 0;
    bool handled = false;
    int rc = acquire_four_step(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, P.res_carr, P.res_cph, P.res_met, P.res_fb,
                               P.res_fi, &handled, true);
    if (handled && P.mode == 1) return rc;          // queued; nothing has been looked at
    if (!handled)
        rc = acquire_passes(c, x, n_samples, prn0, n_prn, n_blocks, noncoh, P.res_carr, P.res_cph, P.res_met, P.res_fb, P.res_fi);
    // (a path without the device-led sequence: it ran eagerly; sgx_acquire_end hands its outputs over)
    P.mode = 2;
    P.n_prn = n_prn;
    P.rc = rc;
    return SGX_OK;
}

extern "C" int sgx_acquire_end(sgx_ctx* c, double* carrFreq, double* codePhase, double* peakMetric, int32_t* freqBin,
                               int32_t* fineIdx) {
    SGX_CHECK_ARG(c && carrFreq && codePhase && peakMetric && freqBin && fineIdx);
    SGX_HIP(hipSetDevice(c->device));
    return sgx_acquire_finish(c, carrFreq, codePhase, peakMetric, freqBin, fineIdx);
}


// ================================ round 6: the sharded search as ONE call ================================
// BASELINE configs[3]: the PRN loop (acquisition.py:92) shards over the ranks, the peaks are gathered.  Rounds 1-5 did the
// pack, the gather and the merge in Python around sgx_acquire (softgnss-python_amd/shard.py): 0.17-0.28 ms of host time per
// call next to a 0.45 ms shard.  Here the rank's search is queued, its peaks are packed into 40-byte records ON THE DEVICE
// behind it, one ncclAllGather follows on the same stream, a small kernel copies the gathered records to the result page and
// the host looks ONCE; the merge into the 32-entry arrays is a loop over at most 32 records.
struct PeakRec {      // = shard.PEAK_DTYPE, 40 bytes
    int prn0, freqBin;
    double carrFreq, codePhase, peakMetric;
    int fineIdx, valid;   // valid 1; 0 unused slot; -1 the reference's IndexError at this PRN; -2 its fine window leaves the record
};
static_assert(sizeof(PeakRec) == 40, "shard.PEAK_DTYPE");

__global__ __launch_bounds__(64) void acq_pack_kernel(const CoarseLook* __restrict__ stage, const long long* __restrict__ fine_bi,
                                                      const int* __restrict__ prn_list, int n_prn, double fs, double npts,
                                                      PeakRec* __restrict__ out, int slots) {
    const int t = threadIdx.x;
    if (t >= slots) return;
    PeakRec r;
    r.prn0 = 0; r.freqBin = -1; r.carrFreq = 0.0; r.codePhase = 0.0; r.peakMetric = 0.0; r.fineIdx = -1; r.valid = 0;
    if (t < n_prn) {
        r.prn0 = prn_list[t];
        r.freqBin = stage->po.fbi[t];
        r.peakMetric = stage->po.peak[t] / stage->second[t];
        r.valid = stage->po.index_error[t] ? -1 : 1;
        if (stage->range_error == 1 + t) r.valid = -2;
        if (r.valid < 0) r.codePhase = (double)stage->po.cph[t];   // (for the error text)
        const int n_det = stage->n_det;
        for (int d = 0; d < n_det; ++d)
            if (stage->det_slot[d] == t && r.valid == 1 && stage->range_error == 0) {
                const long long m = fine_bi[d] - 4;
                r.carrFreq = ((double)m * fs) / npts;
                r.codePhase = (double)stage->det_phase[d];
                r.fineIdx = (int)m;
            }
    }
    out[t] = r;
}

// gathered records -> the result page (its upper half), then the word the host spins on
__global__ __launch_bounds__(256) void acq_gather_publish_kernel(const int* __restrict__ src, int n_words, int* __restrict__ dst,
                                                                 unsigned long long* __restrict__ word, unsigned long long seq) {
    for (int i = threadIdx.x; i < n_words; i += 256) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int sgx_acquire_sharded(sgx_ctx* c, sgx_comm* comm, int32_t rank, int32_t world, const sgx_if* r, size_t offset,
                                   size_t n_samples, int32_t n_prn_total, int32_t n_blocks, int32_t noncoh, double* carrFreq,
                                   double* codePhase, double* peakMetric, int32_t* freqBin, int32_t* fineIdx) {
    SGX_CHECK_ARG(c && r && carrFreq && codePhase && peakMetric && freqBin && fineIdx);
    SGX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world && n_prn_total >= 1 && n_prn_total <= 32);
    SGX_CHECK_ARG(!comm || (comm->n_ranks == world && comm->rank == rank && comm->ctx == c));
    for (int i = 0; i < 32; ++i) {
        carrFreq[i] = 0.0;
        codePhase[i] = 0.0;
        peakMetric[i] = 0.0;
        freqBin[i] = -1;
        fineIdx[i] = -1;
    }
    // contiguous balanced partition (shard.plan_shards)
    const int base = n_prn_total / world, extra = n_prn_total % world;
    const int first = rank * base + (rank < extra ? rank : extra);
    const int n_mine = base + (rank < extra ? 1 : 0);
    const int slots = (n_prn_total + world - 1) / world;
    int32_t prn0[32];
    for (int i = 0; i < n_mine; ++i) prn0[i] = first + i;
    SGX_HIP(hipSetDevice(c->device));
    const size_t rec_bytes = sizeof(PeakRec) * (size_t)slots;
    // where the packed records go: the communicator's send buffer, or (no communicator: one rank, or a shard run alone)
    // the context's small device area
    char* dsm = (char*)c->d_small;
    PeakRec* d_send = comm ? (PeakRec*)comm->d_send : (PeakRec*)(dsm + 720000);
    const PeakRec* d_all = comm ? (const PeakRec*)comm->d_recv : d_send;
    const int n_ranks_seen = comm ? world : 1;
    std::vector<PeakRec> host_pack;      // a search that could not be queued: packed on the host
    bool queued = false;
    if (n_mine > 0) {
        const int rb = sgx_acquire_begin(c, r, offset, n_samples, prn0, n_mine, n_blocks, noncoh);
        if (rb != SGX_OK) return rb;
        queued = c->acq_pending.mode == 1;
        if (!queued) {
            double cf[32], cp[32], pm[32];
            int fb[32], fi[32];
            const int re = sgx_acquire_finish(c, cf, cp, pm, fb, fi);
            if (re != SGX_OK && re != SGX_E_INDEX && re != SGX_E_RANGE) return re;
            host_pack.resize((size_t)slots);
            memset(host_pack.data(), 0, rec_bytes);
            for (int i = 0; i < n_mine; ++i) {
                PeakRec& q = host_pack[(size_t)i];
                q.prn0 = prn0[i]; q.freqBin = fb[i]; q.carrFreq = cf[i]; q.codePhase = cp[i]; q.peakMetric = pm[i];
                q.fineIdx = fi[i]; q.valid = 1;
            }
            if (re != SGX_OK) host_pack[0].valid = re == SGX_E_INDEX ? -1 : -2;   // (every rank learns of it)
        }
    }
    hipStream_t st = c->stream;
    if (queued) {
        const AcqPending& P = c->acq_pending;
        acq_pack_kernel<<<1, 64, 0, st>>>((const CoarseLook*)(dsm + 700000), ((const CoarseLook*)c->d_look)->fine_bi,
                                          (const int*)(dsm + 64), P.n_prn, c->s.samplingFreq, (double)P.npts, d_send, slots);
    } else {
        if (host_pack.empty()) {
            host_pack.resize((size_t)slots);
            memset(host_pack.data(), 0, rec_bytes);
        }
        SGX_HIP(hipMemcpyAsync(d_send, host_pack.data(), rec_bytes, hipMemcpyHostToDevice, st));
    }
    if (comm) {
        const int rg = sgx_comm_allgather_device(comm, rec_bytes);
        if (rg != SGX_OK) return rg;
    }
    const size_t all_bytes = rec_bytes * (size_t)n_ranks_seen;
    if (all_bytes > SGX_TRK_LOOK_OFFSET - SGX_GATHER_LOOK_OFFSET - 16) {
        sgx_set_error("sgx_acquire_sharded: %d ranks x %d slots do not fit the result page", world, slots);
        return SGX_E_ARG;
    }
    const unsigned long long seq = ++c->look_seq;
    char* page_d = (char*)c->d_look + SGX_GATHER_LOOK_OFFSET;
    const char* page_h = (const char*)c->h_look + SGX_GATHER_LOOK_OFFSET;
    acq_gather_publish_kernel<<<1, 256, 0, st>>>((const int*)d_all, (int)(all_bytes / 4), (int*)(page_d + 16),
                                                 (unsigned long long*)page_d, seq);
    SGX_HIP(hipGetLastError());
    {   // the one look
        const unsigned long long* word = (const unsigned long long*)page_h;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned it = 0; !seen; ++it) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) seen = true;
            else if ((it & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) break;
        }
        if (!seen) {
            SGX_HIP(hipStreamSynchronize(st));
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
                sgx_set_error("sharded acquisition: the gathered peaks were not written");
                return SGX_E_HIP;
            }
        }
    }
    if (queued) {   // (device time of this rank's search; the search's own page is complete: the gather came behind it)
        c->acq_pending.mode = 0;
        SGX_HIP(hipEventSynchronize(c->ev[2]));
        hipEventElapsedTime(&c->timing.acquire_ms, c->ev[0], c->ev[2]);
        c->timing.acq_coarse_ms = __builtin_nanf("");
        c->timing.acq_fine_ms = __builtin_nanf("");
    }
    const PeakRec* all = (const PeakRec*)(page_h + 16);
    const int n_rec = slots * n_ranks_seen;
    for (int i = 0; i < n_rec; ++i) {
        const PeakRec& q = all[i];
        if (q.valid == 0) continue;
        if (q.valid == -1) {
            sgx_set_error("IndexError: index %lld is out of bounds for axis 1 with size %lld "
                          "(PRN index %d, codePhase %d; reference acquisition.py:152-162)",
                          (long long)c->n_code, (long long)c->n_code, q.prn0, (int)q.codePhase);
            return SGX_E_INDEX;
        }
        if (q.valid == -2) {
            sgx_set_error("fine search needs codePhase + 10 ms = %lld samples, record window has %zu "
                          "(reference acquisition.py:177 would fail to broadcast)", (long long)q.codePhase + 10 * c->n_code, n_samples);
            return SGX_E_RANGE;
        }
        if (q.prn0 < 0 || q.prn0 >= 32) continue;
        carrFreq[q.prn0] = q.carrFreq;
        codePhase[q.prn0] = q.codePhase;
        peakMetric[q.prn0] = q.peakMetric;
        freqBin[q.prn0] = q.freqBin;
        fineIdx[q.prn0] = q.fineIdx;
    }
    return SGX_OK;
}
