// Tracking of float32 records (Settings.dataType 'float32'; reference tracking.py:154 reads np.fromfile(fid, dataType,
// blksize) and computes in float64).  The fast way is EXACT NARROWING: when every sample of the window is m 2^-k for one k
// (of either sign) and integers m that fit 8 or 16 bits - floats written from ADC samples (k = 0) or normalised by a power
// of two (int16 / 32768: k = 15) are - the integers are tracked by the int8 / int16 kernels; a power of two commutes with
// every rounding of the reference's arithmetic, so the correlator series are the integer record's times 2^-k exactly and
// everything the discriminators make of them (ratios) is untouched.  A record of arbitrary floats has no such k, and a
// channel may start inside a sample of the file (the reference seeks skipNumberOfBytes + codePhase BYTES, tracking.py:107):
// both go to the per-sample kernel of sgx_trk_any.hip, which reads every float where it lies.
#include "sgx_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

// sgx_trk.hip
int sgx_track_kind(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch,
                   int32_t ms, double* out, int32_t* ms_done, int kind, long long skip_bytes, double fscale);

// the power of two that brings a record whose largest |sample| is mx to at most 128 (what the typed kernels' fixed point
// is cut for); 1 for an all-zero record
// The typed kernel's granules are a fixed point cut for samples of comparable size (2^-28 of the largest): a record whose
// largest sample towers 2^12 times above the mean |x| (an outlier, a burst) would lose the small ones' last digits against
// the reference's float64 sums - such a record takes the per-sample kernel (0 is returned).
static double scale_for(double mx, double sum_abs, long long n) {
    if (!(mx > 0.0)) return 1.0;
    if (n > 0 && mx > 4096.0 * (sum_abs / (double)n)) return 0.0;
    int ex = 0;
    (void)frexp(mx, &ex);          // mx = m 2^ex, 1/2 <= m < 1
    // (records near the ends of the float64 range: the scale, its reciprocal or the reference's own I^2 + Q^2 would
    // leave the range - the per-sample kernel follows the reference there, overflow and all)
    if (7 - ex > 900 || 7 - ex < -900) return 0.0;
    return ldexp(1.0, 7 - ex);
}

// [0]: largest |x| (float bits), [1]: smallest exponent of a sample's lowest set bit + 1024 (0x7FFFFFFF: no nonzero sample),
// [2]: a sample that is not finite was seen; sum_abs: the sum of |x| over the finite samples (for a threshold only: the
// order of its additions is not fixed)
__global__ __launch_bounds__(256) void f32_scan_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ st,
                                                       double* __restrict__ sum_abs) {
    unsigned mx = 0u, lo = 0x7FFFFFFFu, bad = 0u;
    double sa = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(x[i]) & 0x7FFFFFFFu;
        const unsigned e = u >> 23, m = u & 0x7FFFFFu;
        if (e == 255u) bad = 1u;
        if (u != 0u && e != 255u) {
            // value = (m | implicit) 2^(e - 150) (a subnormal: m 2^-149): the exponent of its lowest set bit
            const unsigned full = e ? (m | 0x800000u) : m;
            const int lb = (int)(e ? e : 1u) - 150 + (__ffs((int)full) - 1);
            const unsigned key = (unsigned)(lb + 1024);
            lo = key < lo ? key : lo;
            mx = u > mx ? u : mx;
            sa += (double)__uint_as_float(u);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned omx = __shfl_down(mx, off), olo = __shfl_down(lo, off), ob = __shfl_down(bad, off);
        mx = omx > mx ? omx : mx;
        lo = olo < lo ? olo : lo;
        bad |= ob;
        sa += __shfl_down(sa, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&st[0], mx);
        atomicMin(&st[1], lo);
        if (bad) atomicOr(&st[2], 1u);
        if (sum_abs) atomicAdd(sum_abs, sa);
    }
}

// y[i] = x[i] 2^k as an 8- or 16-bit integer (exact by construction)
template <typename T>
__global__ __launch_bounds__(256) void f32_narrow_kernel(const float* __restrict__ x, long long n, int k, T* __restrict__ y) {
    const float s = ldexpf(1.0f, k);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = (T)(int)(x[i] * s);
}

int sgx_track_float32(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch,
                      int32_t ms, double* out, int32_t* ms_done) {
    SGX_CHECK_ARG(c && r && ch && out && ms_done);
    SGX_CHECK_ARG(n_ch >= 1 && ms >= 1);
    const sgx_settings& S = c->s;
    const long long skip = (long long)S.skipNumberOfBytes;
    // the per-sample kernel (fs = 0), or the latency-mode kernel on samples scaled by the power of two fs
    auto generic = [&](double fs = 0.0) {
        return sgx_track_kind(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done, SGX_DT_FLOAT32, skip, fs);
    };
    const char* ne = getenv("SGX_TRK_F32_NARROW");       // '0': no narrowing - every float32 record as floats
    const bool no_narrow = ne && ne[0] == '0';
    long long first = -1, last = -1;
    for (int i = 0; i < n_ch; ++i) {
        if (ch[i].prn == 0) continue;
        const long long p0 = skip + (long long)ch[i].codePhase - rec_file_offset;
        if (p0 < 0) {
            sgx_set_error("channel %d starts at file byte %lld, before the record (offset %lld)", i,
                          skip + (long long)ch[i].codePhase, (long long)rec_file_offset);
            return SGX_E_RANGE;
        }
        // a channel that starts inside a sample of the file reads the bit patterns of bytes of two samples (the reference
        // does): nothing to narrow
        if (p0 % 4 != 0) return generic();
        first = (first < 0 || p0 < first) ? p0 : first;
        last = p0 > last ? p0 : last;
    }
    if (first < 0) return generic();   // nothing to track
    // the window the channels can reach: the same allowance per block as the kernels' units (64 samples, sgx_trk.hip)
    const long long n_code = c->n_code;
    long long end = last + ((long long)ms * (n_code + 64) + n_code) * 4;
    if (end > (long long)r->n) end = (long long)r->n;
    end &= ~3ll;
    const long long n_samp = (end - first) / 4;
    if (n_samp <= 0) return generic();  // (no whole sample: the kernel reports the short read)
    {
        const int rq = sgx_if_require(r, (size_t)end);   // (a streaming record: the scan needs every sample)
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    unsigned* d_st = nullptr;
    SGX_HIP(hipMalloc((void**)&d_st, 4 * sizeof(unsigned) + sizeof(double)));   // [3 words | pad | sum of |x|]
    const unsigned h_init[6] = {0u, 0x7FFFFFFFu, 0u, 0u, 0u, 0u};
    unsigned h_st[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    const float* x = reinterpret_cast<const float*>(r->d + first);
    hipError_t e = hipMemcpyAsync(d_st, h_init, sizeof(h_init), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int grid = (int)((n_samp + 255) / 256 < 4096 ? (n_samp + 255) / 256 : 4096);
        f32_scan_kernel<<<grid, 256, 0, c->stream>>>(x, n_samp, d_st, reinterpret_cast<double*>(d_st + 4));
        e = hipMemcpyAsync(h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_st);
    if (e != hipSuccess) {
        sgx_set_error("float32 record scan: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    if (h_st[2]) return generic();      // NaN or infinite samples: numpy's arithmetic carries them, so does the kernel
    int k = 0;
    double peak = 0.0;
    if (h_st[1] != 0x7FFFFFFFu) {
        const int lb = (int)h_st[1] - 1024;     // every sample is a multiple of 2^lb
        k = -lb;                                // (negative for records of multiples of 2^j, j > 0)
        float mxf;
        memcpy(&mxf, &h_st[0], sizeof(mxf));
        peak = ldexp((double)mxf, k);
    }
    {
        float mxf;
        memcpy(&mxf, &h_st[0], sizeof(mxf));
        if (no_narrow || k > 120 || k < -120 || peak > 32767.0)   // arbitrary floats: the latency-mode kernel reads them as they are
        {
            double sum_abs;
            memcpy(&sum_abs, &h_st[4], sizeof(sum_abs));
            return generic(scale_for((double)mxf, sum_abs, n_samp));
        }
    }
    const bool narrow8 = peak <= 127.0;
    const int sb = narrow8 ? 1 : 2;
    // the integer record: the window only, its first sample at record byte 0
    sgx_if tmp;
    tmp.device = c->device;
    tmp.n = (size_t)n_samp * (size_t)sb;
    e = hipMalloc((void**)&tmp.d, tmp.n + SGX_IF_PAD);
    if (e != hipSuccess) {
        sgx_set_error("hipMalloc(%zu) for the narrowed float32 record failed: %s", tmp.n + SGX_IF_PAD, hipGetErrorString(e));
        return SGX_E_NOMEM;
    }
    int rc = SGX_OK;
    {
        const int grid = (int)((n_samp + 255) / 256 < 8192 ? (n_samp + 255) / 256 : 8192);
        if (narrow8) f32_narrow_kernel<int8_t><<<grid, 256, 0, c->stream>>>(x, n_samp, k, tmp.d);
        else f32_narrow_kernel<short><<<grid, 256, 0, c->stream>>>(x, n_samp, k, reinterpret_cast<short*>(tmp.d));
        e = hipMemsetAsync(tmp.d + tmp.n, 0, SGX_IF_PAD, c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            sgx_set_error("float32 record narrowing: %s", hipGetErrorString(e));
            rc = SGX_E_HIP;
        }
    }
    if (rc == SGX_OK) {
        // Byte p of the float file is byte (p - F) sb / 4 of the integer record, F = the window's first file byte.  The
        // integer kernels place a channel at skipNumberOfBytes + codePhase - rec_file_offset, so the channels are handed
        // over with skipNumberOfBytes folded into codePhase... which is a double and may not be changed; instead the
        // record offset is chosen so that channel i lands on sample (p0_i - first) / 4: that needs one offset per
        // channel, so the channels are tracked with codePhase' = (p0_i - first) / 4 * sb and skip' = offset' = 0.
        std::vector<sgx_chan_init> hc((size_t)n_ch);
        for (int i = 0; i < n_ch; ++i) {
            hc[(size_t)i] = ch[i];
            if (ch[i].prn == 0) continue;
            const long long p0 = skip + (long long)ch[i].codePhase - rec_file_offset;
            hc[(size_t)i].codePhase = (double)((p0 - first) / 4 * sb);
        }
        rc = sgx_track_kind(c, &tmp, 0, hc.data(), n_ch, ms, out, ms_done, narrow8 ? SGX_DT_INT8 : SGX_DT_INT16, 0, 0.0);
        if (rc == SGX_OK) {
            // absoluteSample is fid.tell() in BYTES of the float file (tracking.py:255): integer-record bytes * 4 / sb
            // behind the window's first byte; the six correlator series carry the 2^k
            const double unscale = ldexp(1.0, -k);
            const double file0 = (double)(rec_file_offset + first);
            for (int i = 0; i < n_ch; ++i) {
                double* o = out + (size_t)i * SGX_NUM_SERIES * (size_t)ms;
                if (ch[i].prn == 0) continue;
                const int done = ms_done[i] < ms ? ms_done[i] : ms;
                for (int t = 0; t < done; ++t) o[t] = o[t] * (4.0 / sb) + file0;
                for (int s = 3; s <= 8; ++s)
                    for (int t = 0; t < done; ++t) o[(size_t)s * ms + t] *= unscale;
            }
        }
    }
    hipFree(tmp.d);
    tmp.d = nullptr;
    return rc;
}

// ---- float64 records: the same latency-mode kernel on samples scaled by a power of two, after one scan of the window -------
// [0]: largest |x| (the bits of a non-negative double order like integers), [1]: a sample that is not finite was seen,
// [2]: the sum of |x| over the finite samples (a double; for a threshold only)
__global__ __launch_bounds__(256) void f64_scan_kernel(const double* __restrict__ x, long long n, unsigned long long* __restrict__ st) {
    unsigned long long mx = 0ull, bad = 0ull;
    double sa = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(x[i]) & 0x7FFFFFFFFFFFFFFFull;
        if ((u >> 52) == 0x7FFull) bad = 1ull;
        else {
            mx = u > mx ? u : mx;
            sa += __longlong_as_double((long long)u);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long omx = __shfl_down(mx, off), ob = __shfl_down(bad, off);
        mx = omx > mx ? omx : mx;
        bad |= ob;
        sa += __shfl_down(sa, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&st[0], mx);
        if (bad) atomicOr(&st[1], 1ull);
        atomicAdd(reinterpret_cast<double*>(&st[2]), sa);
    }
}

int sgx_track_float64(sgx_ctx* c, const sgx_if* r, int64_t rec_file_offset, const sgx_chan_init* ch, int32_t n_ch,
                      int32_t ms, double* out, int32_t* ms_done) {
    SGX_CHECK_ARG(c && r && ch && out && ms_done);
    SGX_CHECK_ARG(n_ch >= 1 && ms >= 1);
    const long long skip = (long long)c->s.skipNumberOfBytes;
    auto generic = [&](double fs) {
        return sgx_track_kind(c, r, rec_file_offset, ch, n_ch, ms, out, ms_done, SGX_DT_FLOAT64, skip, fs);
    };
    long long first = -1, last = -1;
    for (int i = 0; i < n_ch; ++i) {
        if (ch[i].prn == 0) continue;
        const long long p0 = skip + (long long)ch[i].codePhase - rec_file_offset;
        if (p0 < 0 || p0 % 8 != 0) return generic(0.0);   // (before the record: reported there; inside a sample: per-sample kernel)
        first = (first < 0 || p0 < first) ? p0 : first;
        last = p0 > last ? p0 : last;
    }
    if (first < 0) return generic(0.0);
    const long long n_code = c->n_code;
    long long end = last + ((long long)ms * (n_code + 64) + n_code) * 8;
    if (end > (long long)r->n) end = (long long)r->n;
    end &= ~7ll;
    const long long n_samp = (end - first) / 8;
    if (n_samp <= 0) return generic(0.0);
    {
        const int rq = sgx_if_require(r, (size_t)end);
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    unsigned long long* d_st = nullptr;
    SGX_HIP(hipMalloc((void**)&d_st, 3 * sizeof(unsigned long long)));
    unsigned long long h_st[3] = {0ull, 0ull, 0ull};
    hipError_t e = hipMemsetAsync(d_st, 0, sizeof(h_st), c->stream);
    if (e == hipSuccess) {
        const int grid = (int)((n_samp + 255) / 256 < 4096 ? (n_samp + 255) / 256 : 4096);
        f64_scan_kernel<<<grid, 256, 0, c->stream>>>(reinterpret_cast<const double*>(r->d + first), n_samp, d_st);
        e = hipMemcpyAsync(h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_st);
    if (e != hipSuccess) {
        sgx_set_error("float64 record scan: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    if (h_st[1]) return generic(0.0);        // NaN or infinite samples: numpy's arithmetic carries them, so does the per-sample kernel
    double mx, sum_abs;
    memcpy(&mx, &h_st[0], sizeof(mx));
    memcpy(&sum_abs, &h_st[2], sizeof(sum_abs));
    return generic(scale_for(mx, sum_abs, n_samp));
}
