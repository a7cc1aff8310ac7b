// Raw-data statistics of Settings.probeData (reference initialize.py:330-417):
// the step before acquisition.  Welch power spectral density of the first 10 code periods,
//   welch(data - mean(data), fs/1e6, hamming(16384, sym=False), nperseg 16384, noverlap 1024, nfft 16384)
// (scipy defaults: constant detrend per segment, density scaling, one-sided, mean over segments), and the
// histogram np.histogram(data, arange(-128, 128)).  The FFT passes are the acquisition's (sgx_fft.hip).
//
//   probe_hist_kernel     256-bin LDS histogram of the int8 window (also yields the exact sum -> mean)
//   probe_segment_kernel  one workgroup per segment: x = d - mean, minus the segment mean, times the window
//   probe_psd_kernel      per bin: mean over segments of |X|^2 * scale (* 2 inside the band)
#include <math.h>

#include "sgx_internal.h"

#define PROBE_NSEG 16384
#define PROBE_NOVERLAP 1024
#define PROBE_BINS (PROBE_NSEG / 2 + 1)

__global__ __launch_bounds__(256) void probe_hist_kernel(const int8_t* __restrict__ x, long long n,
                                                         unsigned long long* __restrict__ hist) {
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        atomicAdd(&s_h[(int)x[i] + 128], 1u);
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void probe_segment_kernel(const int8_t* __restrict__ x, cplx* __restrict__ out,
                                                            const double* __restrict__ win, double mean, int step) {
    __shared__ double s_part[256];
    const int8_t* __restrict__ seg = x + (long long)blockIdx.x * step;
    double acc = 0.0;
    for (int i = threadIdx.x; i < PROBE_NSEG; i += 256) acc += (double)seg[i] - mean;
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s_part[threadIdx.x] += s_part[threadIdx.x + st];
        __syncthreads();
    }
    const double seg_mean = s_part[0] / (double)PROBE_NSEG;
    cplx* __restrict__ o = out + (long long)blockIdx.x * PROBE_NSEG;
    for (int i = threadIdx.x; i < PROBE_NSEG; i += 256)
        o[i] = make_double2(win[i] * (((double)seg[i] - mean) - seg_mean), 0.0);
}

__global__ __launch_bounds__(256) void probe_psd_kernel(const cplx* __restrict__ spec, double* __restrict__ pxx,
                                                        int n_seg, double scale) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= PROBE_BINS) return;
    const double two = (k >= 1 && k < PROBE_BINS - 1) ? 2.0 : 1.0;
    double acc = 0.0;
    for (int s = 0; s < n_seg; ++s) {
        const cplx v = spec[(long long)s * PROBE_NSEG + k];
        acc += ((v.x * v.x + v.y * v.y) * scale) * two;      // conj(X) * X, *= scale, *= 2
    }
    pxx[k] = acc / (double)n_seg;
}

extern "C" int sgx_probe_stats(sgx_ctx* c, const sgx_if* rec, size_t offset, size_t n, double fs_mhz, double* f,
                               double* pxx, int64_t* hist, int32_t* n_segments) {
    SGX_CHECK_ARG(c && rec && f && pxx && hist && n_segments && fs_mhz > 0.0);
    SGX_CHECK_ARG(rec->device == c->device);
    if (offset > rec->n || n > rec->n - offset) {
        sgx_set_error("probe window [%zu, %zu) outside the %zu-sample record", offset, offset + n, rec->n);
        return SGX_E_RANGE;
    }
    if (n < (size_t)PROBE_NSEG) {
        // scipy falls back to nperseg = len(x) with a warning and then rejects the 16384-point window
        sgx_set_error("ValueError: probeData needs at least %d samples for one Welch segment, got %zu", PROBE_NSEG, n);
        return SGX_E_RANGE;
    }
    {
        const int rq = sgx_if_require(rec, offset + n);
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int step = PROBE_NSEG - PROBE_NOVERLAP;
    const int n_seg = (int)((n - PROBE_NOVERLAP) / (size_t)step);
    const int8_t* x = rec->d + offset;

    // window and scale on the host: hamming(16384, sym=False) = 0.54 - 0.46 cos(2 pi k / 16384)
    std::vector<double> win((size_t)PROBE_NSEG);
    double w2 = 0.0;
    for (int k = 0; k < PROBE_NSEG; ++k) {
        win[(size_t)k] = 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)PROBE_NSEG);
        w2 += win[(size_t)k] * win[(size_t)k];
    }
    const double scale = 1.0 / (fs_mhz * w2);

    int rc = sgx_fft_plan_create(&c->plan_probe, PROBE_NSEG);   // 16384 = 16 * 16 * 16 * 4, kept with the context
    if (rc != SGX_OK) return rc;

    const size_t row_bytes = sizeof(cplx) * (size_t)PROBE_NSEG;
    char* d_all = nullptr;
    const size_t bytes = 2 * (size_t)n_seg * row_bytes + sizeof(double) * (size_t)PROBE_NSEG +
                         sizeof(double) * (size_t)PROBE_BINS + 256 * sizeof(unsigned long long);
    SGX_HIP(hipMalloc((void**)&d_all, bytes));
    cplx* d_a = (cplx*)d_all;
    cplx* d_b = d_a + (size_t)n_seg * PROBE_NSEG;
    double* d_win = (double*)(d_b + (size_t)n_seg * PROBE_NSEG);
    double* d_pxx = d_win + PROBE_NSEG;
    unsigned long long* d_hist = (unsigned long long*)(d_pxx + PROBE_BINS);
    unsigned long long h_hist[256];
    hipError_t e = hipMemsetAsync(d_hist, 0, sizeof(h_hist), st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_win, win.data(), sizeof(double) * (size_t)PROBE_NSEG, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipEventRecord(c->ev[0], st);
        probe_hist_kernel<<<256, 256, 0, st>>>(x, (long long)n, d_hist);
        e = hipMemcpyAsync(h_hist, d_hist, sizeof(h_hist), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        hipFree(d_all);
        sgx_set_error("probe histogram failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    long long sum = 0;
    for (int v = 0; v < 256; ++v) sum += (long long)(v - 128) * (long long)h_hist[v];
    const double mean = (double)sum / (double)n;              // np.mean of an int8 array: exact sum, one division
    for (int b = 0; b < 255; ++b) hist[b] = (int64_t)h_hist[b];
    hist[254] += (int64_t)h_hist[255];                        // np.histogram's last bin [126, 127] is closed

    probe_segment_kernel<<<n_seg, 256, 0, st>>>(x, d_a, d_win, mean, step);
    cplx* res = nullptr;
    rc = sgx_fft_forward(&c->plan_probe, d_a, d_b, n_seg, st, &res, PROBE_NSEG);
    if (rc != SGX_OK) {
        hipFree(d_all);
        return rc;
    }
    probe_psd_kernel<<<(PROBE_BINS + 255) / 256, 256, 0, st>>>(res, d_pxx, n_seg, scale);
    hipEventRecord(c->ev[1], st);
    e = hipMemcpyAsync(pxx, d_pxx, sizeof(double) * (size_t)PROBE_BINS, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    hipFree(d_all);
    if (e != hipSuccess) {
        sgx_set_error("probe spectrum failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    // np.fft.rfftfreq(16384, 1 / fs): val = 1 / (n d); f = arange(n/2 + 1) * val
    const double d = 1.0 / fs_mhz;
    const double val = 1.0 / ((double)PROBE_NSEG * d);
    for (int k = 0; k < PROBE_BINS; ++k) f[k] = (double)k * val;
    *n_segments = n_seg;
    return SGX_OK;
}
