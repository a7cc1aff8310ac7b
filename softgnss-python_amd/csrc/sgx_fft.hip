// Batched complex128 forward DFT for gfx950: Stockham autosort passes through HBM/L2, mixed radix.
//
// samplesPerCode = 38192 = 16*7*11*31 is not a power of two (SURVEY.md section 7 hard part 4), the
// fine-frequency search needs 2^22 points.  Each pass is one launch over [rows][n/R] butterflies:
//   thread j: k = j mod Ns; v[q] = in[j + q*n/R] * W_n^(q*k*n/(Ns*R)); V = DFT_R(v);
//             out[(j/Ns)*Ns*R + k + q*Ns] = V[q]
// Radix 16/8/4/2 are register butterflies built from radix-4/2; odd radices (3..31) use a direct
// DFT that pairs v[q] +- v[R-q] (conjugate symmetry of the roots: (R-1) FMAs per output instead
// of 4(R-1)).  fp64 throughout: the acquisition argmax indices must match the reference bit for
// bit, and the fine-search top-2 bins can differ by 4e-5 relative.
// MFMA is deliberately unused (BASELINE.json north_star).
#include <math.h>

#include <mutex>

#include "sgx_internal.h"

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward DFT quarter turn)
__device__ __forceinline__ cplx mul_mi(cplx a) { return make_double2(a.y, -a.x); }

__device__ __forceinline__ void bfly4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    const cplx t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi(csub(x1, x3));
    x0 = cadd(t0, t2);
    x1 = cadd(t1, t3);
    x2 = csub(t0, t2);
    x3 = csub(t1, t3);
}

struct PassArgs {
    const cplx* in;
    cplx* out;
    const cplx* tw_hi;
    const cplx* tw_lo;
    const cplx* wr;        // roots of unity of the radix: wr[m] = exp(-2 pi i m / R)
    long long n;
    long long ns;
    long long nonzero_len;  // input elements >= this index are zero (first pass of a padded row)
    int lo_bits;
    // MODE 1 (first pass): input row r is conj(mul_x[bk]) * mul_f[prn] instead of in[r]
    const cplx* mul_x;
    const cplx* mul_f;
    const int2* row_map;    // (bk, prn) per row, or null: bk = r % rows_per_prn, prn = prn_base + r / rows_per_prn
    int rows_per_prn;
    int prn_base;
    // MODE 2 (last pass): nothing is stored; |V|^2 * inv_n^2 is reduced to one (max, first index) per workgroup
    double* pmax;           // [rows][gridDim.x]
    int* parg;
    double inv_n;
};

__device__ __forceinline__ cplx twiddle(const PassArgs& a, long long t) {
    const cplx h = a.tw_hi[t >> a.lo_bits];
    const cplx l = a.tw_lo[t & ((1ll << a.lo_bits) - 1)];
    return cmul(h, l);
}

// Odd radix: out[k] = v0 + sum_q (a_q c_qk + (b_q.im, -b_q.re) s_qk), out[R-k] with -s, a_q = v[q] + v[R-q],
// b_q = v[q] - v[R-q].  Outputs are handed to `emit` as they are produced (stored, or squared and max-reduced), so
// only a, b and four accumulators are live.  Only the roots m = 1..H are touched (cos is even, sin odd in
// m -> R-m, the sign is a free operand modifier): 4H uniform dwords stay in SGPRs.  With all R-1 roots the
// radix-31 kernel spilled its scalar registers into VGPR lanes and spent two thirds of its instructions on
// v_readlane.
template <int R, class Emit>
__device__ __forceinline__ void dft_odd_emit(const cplx (&v)[R], const cplx* __restrict__ wr, Emit&& emit) {
    static_assert(R % 2 == 1, "odd radix expected");
    constexpr int H = (R - 1) / 2;
    cplx a[H], b[H];
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        a[q - 1] = cadd(v[q], v[R - q]);
        b[q - 1] = csub(v[q], v[R - q]);
    }
    const cplx v0 = v[0];
    cplx s0 = v0;
#pragma unroll
    for (int q = 0; q < H; ++q) s0 = cadd(s0, a[q]);
    double wc[H], ws[H];   // cos(2 pi m / R), -sin(2 pi m / R), m = 1..H
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        wc[m - 1] = wr[m].x;
        ws[m - 1] = wr[m].y;
    }
    emit(0, s0);
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        double pr = v0.x, pi = v0.y, qr = 0.0, qi = 0.0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            const int m0 = (q * k) % R;
            const bool lowhalf = m0 <= H;
            const int m = lowhalf ? m0 : R - m0;
            const double c = wc[m - 1];
            const double sy = lowhalf ? ws[m - 1] : -ws[m - 1];   // imaginary part of the root m0
            pr = __builtin_fma(a[q - 1].x, c, pr);
            pi = __builtin_fma(a[q - 1].y, c, pi);
            qr = __builtin_fma(b[q - 1].y, -sy, qr);
            qi = __builtin_fma(b[q - 1].x, sy, qi);
        }
        emit(k, make_double2(pr + qr, pi + qi));
        emit(R - k, make_double2(pr - qr, pi - qi));
    }
}

template <int R>
__device__ __forceinline__ void dft_small(cplx (&v)[R], const cplx* __restrict__ wr) {
    if constexpr (R == 2) {
        const cplx a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        bfly4(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {
        // n = n1 + 2 n2 (n1 in 0..1, n2 in 0..3), k = 4 k1 + k2
        cplx y[2][4];
#pragma unroll
        for (int n1 = 0; n1 < 2; ++n1) {
            cplx a0 = v[n1], a1 = v[n1 + 2], a2 = v[n1 + 4], a3 = v[n1 + 6];
            bfly4(a0, a1, a2, a3);
            y[n1][0] = a0;
            y[n1][1] = a1;
            y[n1][2] = a2;
            y[n1][3] = a3;
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const cplx b1 = (k2 == 0) ? y[1][k2] : cmul(y[1][k2], wr[k2]);
            v[k2] = cadd(y[0][k2], b1);
            v[k2 + 4] = csub(y[0][k2], b1);
        }
    } else if constexpr (R == 16) {
        // n = n1 + 4 n2, k = 4 k1 + k2: DFT4 over n2, twiddle W16^(n1 k2), DFT4 over n1
        cplx y[4][4];
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) {
            cplx a0 = v[n1], a1 = v[n1 + 4], a2 = v[n1 + 8], a3 = v[n1 + 12];
            bfly4(a0, a1, a2, a3);
            y[n1][0] = a0;
            y[n1][1] = a1;
            y[n1][2] = a2;
            y[n1][3] = a3;
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            cplx b0 = y[0][k2];
            cplx b1 = (k2 == 0) ? y[1][k2] : cmul(y[1][k2], wr[(1 * k2) & 15]);
            cplx b2 = (k2 == 0) ? y[2][k2] : cmul(y[2][k2], wr[(2 * k2) & 15]);
            cplx b3 = (k2 == 0) ? y[3][k2] : cmul(y[3][k2], wr[(3 * k2) & 15]);
            bfly4(b0, b1, b2, b3);
            v[k2] = b0;
            v[k2 + 4] = b1;
            v[k2 + 8] = b2;
            v[k2 + 12] = b3;
        }
    } else {
        dft_odd_emit<R>(v, wr, [&](int q, cplx V) { v[q] = V; });
    }
}

// MODE 0: plain pass.  MODE 1: first pass of the correlation batch, the pointwise product conj(X) * F is formed
// on load (no separate multiply kernel, no product buffer).  MODE 2: last pass of the correlation batch, the
// outputs are squared, scaled and max-reduced in place (no output rows, no separate power kernel).
template <int R, int TPB, int MODE>
__global__ __launch_bounds__(TPB) void fft_pass_kernel(PassArgs a) {
    const long long m = a.n / R;
    const long long j = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long row = blockIdx.y;
    const bool live = j < m;
    if (MODE != 2 && !live) return;
    const cplx* __restrict__ in = a.in + row * a.n;
    cplx* __restrict__ out = a.out + row * a.n;
    const long long k = j % a.ns;
    cplx v[R];
    if (MODE == 1) {
        int bk, prn;
        if (a.row_map) {
            const int2 rm = a.row_map[row];
            bk = rm.x;
            prn = rm.y;
        } else {
            bk = (int)(row % a.rows_per_prn);
            prn = a.prn_base + (int)(row / a.rows_per_prn);
        }
        const cplx* __restrict__ px = a.mul_x + (long long)bk * a.n;
        const cplx* __restrict__ pf = a.mul_f + (long long)prn * a.n;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long long idx = j + q * m;
            const cplx xv = px[idx], fv = pf[idx];
            v[q] = make_double2(__builtin_fma(xv.x, fv.x, xv.y * fv.y), __builtin_fma(xv.x, fv.y, -(xv.y * fv.x)));
        }
    } else {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long long idx = j + q * m;
            v[q] = (live && idx < a.nonzero_len) ? in[idx] : make_double2(0.0, 0.0);
        }
    }
    if (a.ns > 1) {
        const long long tstep = k * (a.n / (a.ns * R));
        long long t = 0;
#pragma unroll
        for (int q = 1; q < R; ++q) {
            t += tstep;
            if (t >= a.n) t -= a.n;
            v[q] = cmul(v[q], twiddle(a, t));
        }
    }
    const long long j0 = (j / a.ns) * a.ns * R + k;
    constexpr bool kEmit = (R % 2 == 1) && (R >= 11);   // large odd radix: consume outputs as they are produced
    if (MODE == 2) {
        // acquisition.py:124-126 abs(ifft(.))**2 for this thread's outputs, then (max, FIRST index)
        double best = -1.0;
        int arg = 0;
        auto take = [&](int q, cplx V) {
            const double re = V.x * a.inv_n, im = V.y * a.inv_n;
            const double pw = re * re + im * im;
            const int idx = (int)(j0 + q * a.ns);
            if (pw > best || (pw == best && idx < arg)) {
                best = pw;
                arg = idx;
            }
        };
        if (live) {
            if constexpr (kEmit) {
                dft_odd_emit<R>(v, a.wr, take);
            } else {
                dft_small<R>(v, a.wr);
#pragma unroll
                for (int q = 0; q < R; ++q) take(q, v[q]);
            }
        }
        __shared__ double s_v[TPB];
        __shared__ int s_i[TPB];
        s_v[threadIdx.x] = best;
        s_i[threadIdx.x] = arg;
        __syncthreads();
        for (int st = TPB / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                const double ov = s_v[threadIdx.x + st];
                const int oi = s_i[threadIdx.x + st];
                if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) {
                    s_v[threadIdx.x] = ov;
                    s_i[threadIdx.x] = oi;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            a.pmax[row * gridDim.x + blockIdx.x] = s_v[0];
            a.parg[row * gridDim.x + blockIdx.x] = s_i[0];
        }
        return;
    }
    if constexpr (kEmit) {
        dft_odd_emit<R>(v, a.wr, [&](int q, cplx V) { out[j0 + q * a.ns] = V; });
    } else {
        dft_small<R>(v, a.wr);
#pragma unroll
        for (int q = 0; q < R; ++q) out[j0 + q * a.ns] = v[q];
    }
}

// ---- plan -----------------------------------------------------------------------------------

static const int kRadixList[] = {16, 8, 4, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31};
#define SGX_MAX_DEVICES 16
static cplx* g_wr[SGX_MAX_DEVICES][32] = {{nullptr}};   // per-device, per-radix root tables

static std::mutex g_wr_lock;   // contexts of several host threads share the tables

static int ensure_roots(int R) {
    std::lock_guard<std::mutex> hold(g_wr_lock);
    int dev = 0;
    SGX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= SGX_MAX_DEVICES) {
        sgx_set_error("device index %d not supported (max %d)", dev, SGX_MAX_DEVICES - 1);
        return SGX_E_ARG;
    }
    if (g_wr[dev][R]) return SGX_OK;
    std::vector<cplx> w((size_t)R);
    for (int m = 0; m < R; ++m) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)R;
        w[(size_t)m] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    SGX_HIP(hipMalloc((void**)&g_wr[dev][R], sizeof(cplx) * (size_t)R));
    SGX_HIP(hipMemcpy(g_wr[dev][R], w.data(), sizeof(cplx) * (size_t)R, hipMemcpyHostToDevice));
    return SGX_OK;
}

int sgx_fft_plan_create(FftPlan* p, int64_t n) {
    if (p->n == n && p->tw_hi) return SGX_OK;
    sgx_fft_plan_destroy(p);
    if (n < 2) {
        sgx_set_error("FFT length %lld not supported", (long long)n);
        return SGX_E_ARG;
    }
    int64_t rem = n;
    p->radices.clear();
    std::vector<int> odd;
    for (int r : kRadixList) {
        while (rem % r == 0) {
            if (r % 2 == 0)
                p->radices.push_back(r);
            else
                odd.push_back(r);
            rem /= r;
        }
    }
    if (rem != 1) {
        sgx_set_error("FFT length %lld has a prime factor above 31 (samplesPerCode must factor into 2..31)",
                      (long long)n);
        return SGX_E_ARG;
    }
    for (int r : odd) p->radices.push_back(r);
    for (int r : p->radices) {
        int rc = ensure_roots(r);
        if (rc != SGX_OK) return rc;
    }
    p->lo_bits = (n > (1 << 18)) ? 11 : 8;
    const int64_t lo_n = 1ll << p->lo_bits;
    const int64_t hi_n = (n + lo_n - 1) / lo_n + 1;
    std::vector<cplx> lo((size_t)lo_n), hi((size_t)hi_n);
    const long double twopi = 2.0L * 3.14159265358979323846264338327950288L;
    for (int64_t t = 0; t < lo_n; ++t) {
        const long double ang = -twopi * (long double)t / (long double)n;
        lo[(size_t)t] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    for (int64_t h = 0; h < hi_n; ++h) {
        const long double ang = -twopi * (long double)((h << p->lo_bits) % n) / (long double)n;
        hi[(size_t)h] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    SGX_HIP(hipMalloc((void**)&p->tw_lo, sizeof(cplx) * (size_t)lo_n));
    SGX_HIP(hipMalloc((void**)&p->tw_hi, sizeof(cplx) * (size_t)hi_n));
    SGX_HIP(hipMemcpy(p->tw_lo, lo.data(), sizeof(cplx) * (size_t)lo_n, hipMemcpyHostToDevice));
    SGX_HIP(hipMemcpy(p->tw_hi, hi.data(), sizeof(cplx) * (size_t)hi_n, hipMemcpyHostToDevice));
    p->n = n;
    return SGX_OK;
}

void sgx_fft_plan_destroy(FftPlan* p) {
    if (p->tw_hi) hipFree(p->tw_hi);
    if (p->tw_lo) hipFree(p->tw_lo);
    p->tw_hi = p->tw_lo = nullptr;
    p->n = 0;
    p->radices.clear();
}

template <int R, int TPB>
static void launch_pass(const PassArgs& a, int64_t rows, hipStream_t st, int mode) {
    const long long m = a.n / R;
    dim3 grid((unsigned)((m + TPB - 1) / TPB), (unsigned)rows);
    if (mode == 1)
        fft_pass_kernel<R, TPB, 1><<<grid, TPB, 0, st>>>(a);
    else if (mode == 2)
        fft_pass_kernel<R, TPB, 2><<<grid, TPB, 0, st>>>(a);
    else
        fft_pass_kernel<R, TPB, 0><<<grid, TPB, 0, st>>>(a);
}

int sgx_fft_last_pass_blocks(const FftPlan* p) {
    const int r = p->radices.back();
    const int tpb = (r == 16 || (r >= 11 && r <= 19)) ? 128 : (r >= 23 ? 64 : 256);
    const long long m = p->n / r;
    return (int)((m + tpb - 1) / tpb);
}

int sgx_fft_forward(const FftPlan* p, cplx* a, cplx* b, int64_t rows, hipStream_t st, cplx** result,
                    int64_t nonzero_len) {
    return sgx_fft_forward_fused(p, a, b, rows, st, result, nonzero_len, nullptr);
}

int sgx_fft_forward_fused(const FftPlan* p, cplx* a, cplx* b, int64_t rows, hipStream_t st, cplx** result,
                          int64_t nonzero_len, const FftFuse* fuse) {
    if (!p->tw_hi || rows < 1 || rows > 65535) {
        sgx_set_error("sgx_fft_forward: bad plan or row count %lld", (long long)rows);
        return SGX_E_ARG;
    }
    int cur_dev = 0;
    SGX_HIP(hipGetDevice(&cur_dev));
    if (cur_dev < 0 || cur_dev >= SGX_MAX_DEVICES) return SGX_E_ARG;
    cplx* src = a;
    cplx* dst = b;
    long long ns = 1;
    bool first = true;
    const size_t n_pass = p->radices.size();
    size_t ipass = 0;
    for (int r : p->radices) {
        const bool last = (++ipass == n_pass);
        int mode = 0;
        PassArgs pa;
        pa.mul_x = pa.mul_f = nullptr;
        pa.row_map = nullptr;
        pa.rows_per_prn = 1;
        pa.prn_base = 0;
        pa.pmax = nullptr;
        pa.parg = nullptr;
        pa.inv_n = 0.0;
        if (fuse && first && fuse->mul_x) {
            mode = 1;
            pa.mul_x = fuse->mul_x;
            pa.mul_f = fuse->mul_f;
            pa.row_map = fuse->row_map;
            pa.rows_per_prn = fuse->rows_per_prn;
            pa.prn_base = fuse->prn_base;
        }
        if (fuse && last && fuse->pmax) {
            if (mode == 1) {
                sgx_set_error("single-pass FFT cannot fuse both ends");
                return SGX_E_ARG;
            }
            mode = 2;
            pa.pmax = fuse->pmax;
            pa.parg = fuse->parg;
            pa.inv_n = fuse->inv_n;
        }
        pa.in = src;
        pa.out = dst;
        pa.tw_hi = p->tw_hi;
        pa.tw_lo = p->tw_lo;
        pa.wr = g_wr[cur_dev][r];
        pa.n = p->n;
        pa.ns = ns;
        pa.nonzero_len = first ? nonzero_len : p->n;
        pa.lo_bits = p->lo_bits;
        switch (r) {
            case 16: launch_pass<16, 128>(pa, rows, st, mode); break;
            case 8: launch_pass<8, 256>(pa, rows, st, mode); break;
            case 4: launch_pass<4, 256>(pa, rows, st, mode); break;
            case 2: launch_pass<2, 256>(pa, rows, st, mode); break;
            case 3: launch_pass<3, 256>(pa, rows, st, mode); break;
            case 5: launch_pass<5, 256>(pa, rows, st, mode); break;
            case 7: launch_pass<7, 256>(pa, rows, st, mode); break;
            case 11: launch_pass<11, 128>(pa, rows, st, mode); break;
            case 13: launch_pass<13, 128>(pa, rows, st, mode); break;
            case 17: launch_pass<17, 128>(pa, rows, st, mode); break;
            case 19: launch_pass<19, 128>(pa, rows, st, mode); break;
            case 23: launch_pass<23, 64>(pa, rows, st, mode); break;
            case 29: launch_pass<29, 64>(pa, rows, st, mode); break;
            case 31: launch_pass<31, 64>(pa, rows, st, mode); break;
            default:
                sgx_set_error("radix %d not instantiated", r);
                return SGX_E_ARG;
        }
        ns *= r;
        first = false;
        cplx* t = src;
        src = dst;
        dst = t;
    }
    SGX_HIP(hipGetLastError());
    *result = src;
    return SGX_OK;
}
