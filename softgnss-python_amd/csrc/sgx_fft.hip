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

#include <atomic>
#include "sgx_internal.h"

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// workgroup barrier that waits for LDS traffic only (__syncthreads() also waits for the wave's global loads - the next
// tile's, requested ahead - and stores)
__device__ __forceinline__ void lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
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

// =====================================================================================================================
// Four-step transform with register-resident sub-transforms (the acquisition's hot transforms).
//
// n = N1 * N2, input index N2 n1 + n2, output index k1 + N1 k2:
//   X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) [ W_n^(n2 k1) sum_n1 x[N2 n1 + n2] W_N1^(n1 k1) ]
//   columns kernel   a workgroup takes 8 adjacent columns n2 (128 contiguous bytes per n1), runs their N1-point
//                    transforms - first radix straight from memory in registers, one exchange through LDS, second radix
//                    in registers - multiplies by W_n^(n2 k1) and stores element (k1, n2) where it came from;
//   rows kernel      a workgroup takes 7 adjacent rows k1 (contiguous elements), runs their N2-point transforms the same
//                    way and either stores X[k1 + N1 k2] or - last step of the correlation - squares, scales and
//                    max-reduces the outputs without storing anything (optionally summing the powers of several 1-ms
//                    blocks first: the non-coherent extension).
// A row therefore crosses HBM twice (write after the columns, read before the rows) instead of once per radix pass.
// The first kernel can form the correlation product conj(X_b[(i + shift) mod n]) * F_prn[i] on load; the circular
// shift is how one forward spectrum serves every Doppler bin that differs from it by whole output bins.

#define F4W_N1 217
#define F4W_N2 176

struct F4Args {
    const cplx* in;
    cplx* out;
    const cplx* tw_hi;
    const cplx* tw_lo;
    const cplx* wr[3];      // root tables of the (up to three) radices of this kernel's sub-transform
    const cplx* tw_sub;     // W_L^t, t < L, of this kernel's sub-transform length L
    long long n;
    long long nonzero_len;  // input elements >= this index are zero
    int lo_bits;
    // product on load (columns kernel, MODE 1)
    const cplx* mul_x;      // forward spectra [block * n_phi + phi][n]
    const cplx* mul_f;      // code spectra [prn][n]
    const int2* bin_map;    // per Doppler bin: (phi index, circular shift)
    const int2* row_map;    // optional: (block * n_bins + bin, prn) per row
    int n_bins, n_phi, rows_per_prn, prn_base;
    int n_blocks, blocks_fast;   // batch rows ordered (prn, bin, block) instead of (prn, block, bin)
    int xcd_order;               // columns kernel, MODE 1: tiles dealt to the XCDs in (PRN, block, tile, bin) order
    // rows kernel, MODE 2 / 3
    double* pmax;           // [rows][gridDim.x] per-workgroup maxima (MODE 2)
    int* parg;
    double* pout;           // [rows][n] powers (MODE 3)
    double inv_n;
    int sum_blocks;         // > 1: powers of this many consecutive input rows are added first (input row = row * sum_blocks + b)
    const int* sec;         // rows kernel, MODE 4: [5][32] row / lo0 / hi0 / lo1 / hi1 of the second-peak search
    double* second_out;     // [32]
    double* t2_b1;          // rows kernel, MODE 5: [rows][217] per output residue: maximum, maximum of the others, first index
    double* t2_b2;
    int* t2_i1;
};

__device__ __forceinline__ cplx f4_twiddle(const F4Args& a, long long t) {
    const cplx h = a.tw_hi[t >> a.lo_bits];
    const cplx l = a.tw_lo[t & ((1ll << a.lo_bits) - 1)];
    return cmul(h, l);
}

// Sub-transform twiddles W_L^t in LDS: one table, or (long transforms) a product of two small ones.
struct TwDirect {
    static constexpr bool kPowers = false;
    const cplx* t;
    __device__ __forceinline__ cplx operator[](int i) const { return t[i]; }
};
struct TwTwoLevel {   // W_L^t = hi[t >> 6] * lo[t & 63]
    // a butterfly's R - 1 twiddles are the powers of ONE: looked up once (two reads and a product), the others by
    // squaring and multiplying (depth 4 for radix 16) - every look-up of this table costs a product anyway, and the
    // 4096-point rows kernel spent more LDS reads on its twiddles than on its data
    static constexpr bool kPowers = true;
    const cplx* hi;
    const cplx* lo;
    __device__ __forceinline__ cplx operator[](int i) const { return cmul(hi[i >> 6], lo[i & 63]); }
};

// w^1 .. w^(R-1) from w by squaring and multiplying (each power is one product of two lower ones, depth log2 R)
template <int R>
__device__ __forceinline__ void twiddle_powers(cplx w, cplx (&p)[R]) {
    p[1] = w;
#pragma unroll
    for (int q = 2; q < R; ++q) p[q] = cmul(p[q / 2], p[q - q / 2]);
}

// One Stockham radix-R pass over C sequences of length L in LDS; element n of sequence c lives at buf[n * SN + c * SC].
// CFAST: consecutive threads take consecutive sequences (use when SC == 1), else consecutive butterflies.
// twl[t] = W_L^t, t < L (needed when NS > 1).
// PADSH > 0 (SN == 1 only): element n lives at n + (n >> PADSH) - one spare element per 2^PADSH, so that a pass whose
// lanes write with a stride of 2^PADSH elements (radix 16, first pass) spreads over the banks; SC is the padded length.
template <int L, int R, int NS, int C, int SN, int SC, int TPB, bool CFAST, int PADSH = 0, class TW>
__device__ __forceinline__ void lds_radix_pass(cplx* __restrict__ buf, const TW twl,
                                               const cplx* __restrict__ wr, int tid) {
    static_assert(PADSH == 0 || SN == 1, "padding is per element");
    constexpr int M = L / R;
    constexpr int TOTAL = M * C;
    constexpr int ROUNDS = (TOTAL + TPB - 1) / TPB;
    constexpr int TS = L / (NS * R);
    auto at = [](int n) { return PADSH ? n + (n >> PADSH) : n; };
    cplx v[ROUNDS][R];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int w = tid + rd * TPB;
        if (w < TOTAL) {
            const int c = CFAST ? w % C : w / M, j = CFAST ? w / C : w % M;
            const int k = j % NS;
            if constexpr (TW::kPowers && (NS > 1)) {
                cplx pw[R];
                twiddle_powers<R>(twl[k * TS], pw);
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    v[rd][q] = buf[at(j + q * M) * SN + c * SC];
                    if (q > 0) v[rd][q] = cmul(v[rd][q], pw[q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    v[rd][q] = buf[at(j + q * M) * SN + c * SC];
                    if (NS > 1 && q > 0) v[rd][q] = cmul(v[rd][q], twl[q * k * TS]);   // q k TS < L
                }
            }
        }
    }
    lds_sync();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int w = tid + rd * TPB;
        if (w < TOTAL) {
            const int c = CFAST ? w % C : w / M, j = CFAST ? w / C : w % M;
            const int k = j % NS;
            const int j0 = (j / NS) * NS * R + k;
            if constexpr ((R % 2 == 1) && (R >= 11)) {
                // large odd radix: outputs go to LDS as they are produced (half the live registers)
                dft_odd_emit<R>(v[rd], wr, [&](int q, cplx V) { buf[at(j0 + q * NS) * SN + c * SC] = V; });
            } else {
                dft_small<R>(v[rd], wr);
#pragma unroll
                for (int q = 0; q < R; ++q) buf[at(j0 + q * NS) * SN + c * SC] = v[rd][q];
            }
        }
    }
    lds_sync();
}

// An odd radix whose outputs are shared by PARTS waves: each produces a compile-time subset of the output pairs
// (k, R - k) from the halves a = v[q] + v[R-q], b = v[q] - v[R-q] (PARTS = 1: all of them).
template <int R, int PART, int PARTS, class Emit>
__device__ __forceinline__ void dft_odd_part(const cplx (&a)[(R - 1) / 2], const cplx (&b)[(R - 1) / 2], cplx v0,
                                             const cplx* __restrict__ wr, Emit&& emit) {
    constexpr int H = (R - 1) / 2;
    double wc[H], ws[H];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        wc[m - 1] = wr[m].x;
        ws[m - 1] = wr[m].y;
    }
    if (PART == 0) {
        cplx s0 = v0;
#pragma unroll
        for (int q = 0; q < H; ++q) s0 = cadd(s0, a[q]);
        emit(0, s0);
    }
#pragma unroll
    for (int k = 1 + PART; k <= H; k += PARTS) {
        double pr = v0.x, pi = v0.y, qr = 0.0, qi = 0.0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            const int m0 = (q * k) % R;
            const bool lowhalf = m0 <= H;
            const int m = lowhalf ? m0 : R - m0;
            const double c = wc[m - 1];
            const double sy = lowhalf ? ws[m - 1] : -ws[m - 1];
            pr = __builtin_fma(a[q - 1].x, c, pr);
            pi = __builtin_fma(a[q - 1].y, c, pi);
            qr = __builtin_fma(b[q - 1].y, -sy, qr);
            qi = __builtin_fma(b[q - 1].x, sy, qi);
        }
        emit(k, make_double2(pr + qr, pi + qi));
        emit(R - k, make_double2(pr - qr, pi - qi));
    }
}

// The columns kernel of the 217 x 176 transform: TWO WAVES per tile of 8 columns.
//   217 = 31 a + b on the way in, c + 7 d on the way out:  X[c + 7 d] = sum_b W_31^(b d) W_217^(b c) sum_a x[31 a + b] W_7^(a c)
//   stage 1   lane (column, b) loads its seven inputs x[31 a + b] straight from memory (forming the correlation product
//             on the way), runs the 7-point transform in registers and writes Y[c][b] W_217^(b c) to LDS: four rounds
//             of 8 b's, two per wave;
//   stage 2   lane (column, c) (56 of 64) reads its 31 values, runs the 31-point transform in registers - the roots in
//             scalar registers, outputs in pairs (d, 31 - d), the odd d in one wave and the even d in the other - and
//             stores each output times W_n^(n2 (c + 7 d)) where it belongs, the factors by two recurrences (up from
//             d = 0, down from d = 31).
// About 2 500 instructions per tile, 27 KB of LDS, 166 registers: five workgroups per CU.  History (config 2's 1 856 rows):
// the round-2 kernel staged the tile through LDS, ran Stockham passes on it and split every radix-31 butterfly over
// four waves (each repeating its input side): 4 800 instructions per tile, bound by instruction issue, 385-415 us.  One
// wave per tile: 337-353 us (five waves per CU leave a SIMD waiting on its own dependent instructions).  This kernel:
// 313-330 us, of which the stores alone account for 287 (measured with the radix-31 arithmetic removed) and the
// arithmetic alone for 230 (measured with the stores removed).
template <int PART>
__device__ __forceinline__ void f4w_stage2(const double (&wc)[15], const double (&wsn)[15], const cplx* __restrict__ y,
                                           cplx w0, cplx ws, cplx wN, cplx* __restrict__ out, unsigned o0) {
    constexpr int C = 8, N2 = F4W_N2, R = 31, H = 15;
    constexpr int NP = PART == 0 ? 8 : 7;   // this wave's output pairs (d, 31 - d): d = 1 + PART, 3 + PART, ...
    const cplx v0 = y[0];
    cplx sa[H], sb[H];
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        const cplx x = y[q * C], z = y[(R - q) * C];
        sa[q - 1] = cadd(x, z);
        sb[q - 1] = csub(x, z);
    }
    if (PART == 0) {
        cplx s0 = v0;
#pragma unroll
        for (int q = 0; q < H; ++q) s0 = cadd(s0, sa[q]);
        out[o0] = cmul(s0, w0);
    }
    // output factors W_n^(n2 (c + 7 d)) by steps of ws^2: up from d = 1 + PART, down from d = 30 - PART
    const cplx ws2 = cmul(ws, ws);
    const cplx first = PART == 0 ? ws : ws2;
    cplx up = cmul(w0, first);
    cplx dn = cmul(cmul(w0, wN), make_double2(first.x, -first.y));
    const cplx ws2c = make_double2(ws2.x, -ws2.y);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int k = 1 + PART + 2 * i;
        double pr = v0.x, pi = v0.y, qr = 0.0, qi = 0.0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            const int m0 = (q * k) % R;
            const bool lowhalf = m0 <= H;
            const int m = lowhalf ? m0 : R - m0;
            const double sy = lowhalf ? wsn[m - 1] : -wsn[m - 1];
            pr = __builtin_fma(sa[q - 1].x, wc[m - 1], pr);
            pi = __builtin_fma(sa[q - 1].y, wc[m - 1], pi);
            qr = __builtin_fma(sb[q - 1].y, -sy, qr);
            qi = __builtin_fma(sb[q - 1].x, sy, qi);
        }
        if (i > 0) {
            up = cmul(up, ws2);
            dn = cmul(dn, ws2c);
        }
        out[o0 + (unsigned)(7 * k * N2)] = cmul(make_double2(pr + qr, pi + qi), up);
        out[o0 + (unsigned)(7 * (R - k) * N2)] = cmul(make_double2(pr - qr, pi - qi), dn);
    }
}

template <int MODE>
__global__ __launch_bounds__(128) void fft4_cols217_kernel(F4Args a) {
    constexpr int N1 = F4W_N1, N2 = F4W_N2, C = 8;
    __shared__ cplx buf[N1 * C];   // [c][b][column]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 7, g = lane >> 3;
    int c0 = blockIdx.x * C;
    long long row = blockIdx.y;
    if (MODE == 1 && a.xcd_order) {
        // The batch in its implicit order: workgroup i runs on XCD i % 8, so XCD x is given the tiles
        // [x T / 8, (x + 1) T / 8) of the order (PRN, block, column tile, Doppler bin) - bins fastest: the workgroups an
        // XCD runs one after the other share the code spectrum's tile and find it in their L2 (in dispatch order the 29
        // bins of a tile land on eight XCDs at unrelated times and the code spectra are re-read through L2 misses:
        // 0.64 GB of the 3.7 GB a config-2 call moved).
        const unsigned id = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
        const unsigned L = (id & 7u) * (total >> 3) + (id >> 3);
        const unsigned bin = L % (unsigned)a.n_bins, t1 = L / (unsigned)a.n_bins;
        const unsigned tile = t1 % gridDim.x, t2 = t1 / gridDim.x;
        const unsigned blk = t2 % (unsigned)a.n_blocks, pr = t2 / (unsigned)a.n_blocks;
        c0 = (int)tile * C;
        row = (long long)pr * a.rows_per_prn + (a.blocks_fast ? bin * a.n_blocks + blk : blk * a.n_bins + bin);
    }
    const int n2 = c0 + col;
    // output factors: W_n^(n2 c), the step W_n^(7 n2) and W_n^(217 n2) (stage 2's lane: c = g)
    const int lo_mask = (1 << a.lo_bits) - 1;
    const int t0 = n2 * (g < 7 ? g : 0), t7 = 7 * n2, t217 = N1 * n2;
    const cplx h0 = a.tw_hi[t0 >> a.lo_bits], l0 = a.tw_lo[t0 & lo_mask];
    const cplx h7 = a.tw_hi[t7 >> a.lo_bits], l7 = a.tw_lo[t7 & lo_mask];
    const cplx hN = a.tw_hi[t217 >> a.lo_bits], lN = a.tw_lo[t217 & lo_mask];
    // the radix-31 roots: read HERE, in front of the barrier and the stores, where the compiler still takes them
    // through the scalar cache into scalar registers (behind those it loads them per lane: sixty vector registers)
    double wc31[15], ws31[15];
    {
        const cplx* __restrict__ wr = a.wr[1];
#pragma unroll
        for (int m = 1; m <= 15; ++m) {
            wc31[m - 1] = wr[m].x;
            ws31[m - 1] = wr[m].y;
        }
    }
    const cplx* __restrict__ in = a.in + row * a.n;
    const cplx* __restrict__ px = nullptr;
    const cplx* __restrict__ pf = nullptr;
    int shift = 0;
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
        const bool bf = a.blocks_fast && !a.row_map;
        const int b = bf ? bk % a.n_blocks : bk / a.n_bins, kb = bf ? bk / a.n_blocks : bk % a.n_bins;
        const int2 bm = a.bin_map[kb];
        px = a.mul_x + (long long)(b * a.n_phi + bm.x) * a.n;
        pf = a.mul_f + (long long)prn * a.n;
        shift = bm.y;
    }
    // ---- stage 1: this wave's rounds are wave and wave + 2 ----
    cplx xv[2][7], fv[2][7], w1[2];
    auto load_round = [&](int r) {
        const int bq = g + 8 * (wave + 2 * r);
        const int b = bq < 31 ? bq : 30;   // (lanes 56..63 of the last round repeat b = 30 and store nothing)
        w1[r] = a.tw_sub[b];
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            // (32-bit element indices on uniform bases: one address add per load)
            const unsigned idx = (unsigned)((31 * m + b) * N2 + n2);
            if (MODE == 1) {
                unsigned ix = idx + (unsigned)shift;
                ix = ix >= (unsigned)(N1 * N2) ? ix - (unsigned)(N1 * N2) : ix;
                xv[r][m] = px[ix];
                fv[r][m] = pf[idx];
            } else {
                xv[r][m] = ((long long)idx < a.nonzero_len) ? in[idx] : make_double2(0.0, 0.0);
            }
        }
    };
    auto run_round = [&](int r) {
        const int b = g + 8 * (wave + 2 * r);
        cplx v[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            v[m] = xv[r][m];
            if (MODE == 1)
                v[m] = make_double2(__builtin_fma(xv[r][m].x, fv[r][m].x, xv[r][m].y * fv[r][m].y),
                                    __builtin_fma(xv[r][m].x, fv[r][m].y, -(xv[r][m].y * fv[r][m].x)));
        }
        cplx tw[7];
        tw[1] = w1[r];
        tw[2] = cmul(tw[1], tw[1]);
        tw[3] = cmul(tw[2], tw[1]);
        tw[4] = cmul(tw[2], tw[2]);
        tw[5] = cmul(tw[4], tw[1]);
        tw[6] = cmul(tw[3], tw[3]);
        cplx sa[3], sb[3];
#pragma unroll
        for (int q = 1; q <= 3; ++q) {
            sa[q - 1] = cadd(v[q], v[7 - q]);
            sb[q - 1] = csub(v[q], v[7 - q]);
        }
        const bool act = b < 31;
        dft_odd_part<7, 0, 1>(sa, sb, v[0], a.wr[0], [&](int c, cplx V) {
            if (c > 0) V = cmul(V, tw[c]);
            if (act) buf[(c * 31 + b) * C + col] = V;
        });
    };
    load_round(0);
    load_round(1);
    run_round(0);
    run_round(1);
    __syncthreads();
    // ---- stage 2 ----
    if (g < 7) {
        const int c = g;
        const cplx* __restrict__ y = buf + (c * 31) * C + col;
        cplx* __restrict__ out = a.out + row * a.n;
        const unsigned o0 = (unsigned)(c * N2 + n2);
        const cplx w0 = cmul(h0, l0), ws = cmul(h7, l7), wN = cmul(hN, lN);
        if (wave == 0) f4w_stage2<0>(wc31, ws31, y, w0, ws, wN, out, o0);
        else f4w_stage2<1>(wc31, ws31, y, w0, ws, wN, out, o0);
    }
}

// (value, first index) maximum over a workgroup of TPB threads; result valid in thread 0.  sv / si: TPB / 64 slots each.
template <int TPB>
__device__ __forceinline__ void wg_argmax(double& best, int& arg, double* __restrict__ sv, int* __restrict__ si, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(arg, off, 64);
        if (ov > best || (ov == best && oi < arg)) {
            best = ov;
            arg = oi;
        }
    }
    if ((tid & 63) == 0) {
        sv[tid >> 6] = best;
        si[tid >> 6] = arg;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < TPB / 64; ++w) {
            const double ov = sv[w];
            const int oi = si[w];
            if (ov > best || (ov == best && oi < arg)) {
                best = ov;
                arg = oi;
            }
        }
    }
}

// The rows kernel of the 217 x 176 transform.
//   176 = a + 16 b on the way in, 11 c + d on the way out:  X[11 c + d] = sum_a W_16^(a c) W_176^(a d) sum_b x[a + 16 b] W_11^(b d)
//   stage 1   lane (row, a) (112 of 128) loads its eleven inputs x[a + 16 b] straight from memory (256 contiguous bytes
//             per row and load), runs the 11-point transform in registers and writes Z[d][a] W_176^(a d) to LDS, 17
//             elements per d (the next stage reads with a stride of one d: no bank conflicts);
//   stage 2   lane (row, d) (77 of 128) reads its 16 values, runs the 16-point transform in registers and squares, adds up
//             (blocks of the non-coherent sum) and max-reduces its outputs where they are.
// 27 LDS accesses per thread, 21 KB of LDS.  The round-2 kernel staged the tile through LDS and ran Stockham passes on it:
// 84 LDS accesses per thread, the radix-16 pass writing with a stride of 16 elements - the counters showed the LDS pipe
// busy for the whole 345 us of config 2, more than a third of that in bank conflicts.  This one: 231-253 us.
#ifndef F4R_WAVES_NB
#define F4R_WAVES_NB 2
#endif
template <int MODE, bool NB1>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(F4R_WAVES_NB))) void fft4_rows176_kernel(F4Args a) {
    constexpr int N1 = F4W_N1, N2 = F4W_N2, CB = 7, DP = 17, RP = 11 * DP;
    __shared__ cplx buf[CB * RP];   // [row][d][a], 17 elements per d
    const int tid = threadIdx.x;
    // roots of both radices: read here, in front of barriers and stores, where they still go through the scalar cache
    cplx w11[11], w16[16];
#pragma unroll
    for (int m = 1; m <= 5; ++m) w11[m] = a.wr[1][m];
#pragma unroll
    for (int m = 1; m < 10; ++m) w16[m] = a.wr[0][m];
    const bool s1 = tid < CB * 16, s2 = tid < CB * 11;
    const int r1 = s1 ? tid >> 4 : CB - 1, a1 = tid & 15;
    const int r2 = s2 ? tid / 11 : 0, d2 = s2 ? tid % 11 : 0;
    // W_176^(a d), d = 1..10, for this lane's a (several blocks per output row: fetched again for every block, the
    // forty registers are wanted by the power accumulators and the next block's loads)
    cplx tw[11];
    if (NB1) {
#pragma unroll
        for (int d = 1; d <= 10; ++d) tw[d] = a.tw_sub[a1 * d];
    }
    const int k10 = blockIdx.x * CB;
    const long long row = blockIdx.y;
    const int nb = (NB1 || MODE == 0 || a.sum_blocks < 1) ? 1 : a.sum_blocks;
    double acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0;
    cplx keep[16];   // MODE 0: the transform itself
    cplx v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10;   // (named: an array inside the block loop is left in scratch memory)
    const unsigned o1 = (unsigned)(r1 * N2 + a1);
#define F4R_LOAD(blk)                                                                              \
    do {                                                                                           \
        const cplx* __restrict__ in = a.in + (row * nb + (blk)) * a.n + (long long)k10 * N2;       \
        v0 = in[o1]; v1 = in[o1 + 16]; v2 = in[o1 + 32]; v3 = in[o1 + 48]; v4 = in[o1 + 64];        \
        v5 = in[o1 + 80]; v6 = in[o1 + 96]; v7 = in[o1 + 112]; v8 = in[o1 + 128]; v9 = in[o1 + 144]; \
        v10 = in[o1 + 160];                                                                        \
    } while (0)
    F4R_LOAD(0);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        if (!NB1) {
            asm volatile("" ::: "memory");   // (keeps these loads inside the loop)
#pragma unroll
            for (int d = 1; d <= 10; ++d) tw[d] = a.tw_sub[a1 * d];
        }
        {
            const cplx v[11] = {v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10};
            cplx* __restrict__ z = buf + r1 * RP + a1;
            dft_odd_emit<11>(v, w11, [&](int d, cplx V) {
                if (d > 0) V = cmul(V, tw[d]);
                if (s1) z[d * DP] = V;
            });
        }
        __syncthreads();
        if (!NB1 && b + 1 < nb) F4R_LOAD(b + 1);   // (in flight behind the second stage)
        {
            cplx y[16];
            const cplx* __restrict__ z = buf + r2 * RP + d2 * DP;
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = z[q];
            dft_small<16>(y, w16);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (MODE == 0) {
                    keep[c] = y[c];
                } else {
                    const double re = y[c].x * a.inv_n, im = y[c].y * a.inv_n;
                    const double pw = re * re + im * im;
                    acc[c] = (b == 0) ? pw : acc[c] + pw;
                }
            }
        }
        if (!NB1 && b + 1 < nb) __syncthreads();
    }
#undef F4R_LOAD
    // output k = k1 + N1 k2, k2 = 11 c + d
    const int idx0 = (k10 + r2) + N1 * d2;
    if (MODE == 0) {
        cplx* __restrict__ out = a.out + row * a.n;
        if (s2) {
#pragma unroll
            for (int c = 0; c < 16; ++c) out[idx0 + N1 * 11 * c] = keep[c];
        }
        return;
    }
    if (MODE == 4) {
        // second-peak search: the maximum over this row's allowed indices, into second_out[row]
        double best = 0.0;
        const int p = (int)row;
        if (s2 && a.sec[p] >= 0) {
            const int lo0 = a.sec[32 + p], hi0 = a.sec[64 + p], lo1 = a.sec[96 + p], hi1 = a.sec[128 + p];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int idx = idx0 + N1 * 11 * c;
                const bool in = (idx >= lo0 && idx < hi0) || (idx >= lo1 && idx < hi1);
                best = in ? fmax(best, acc[c]) : best;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_down(best, off));
        if ((tid & 63) == 0 && best > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(&a.second_out[p]), (unsigned long long)__double_as_longlong(best));
        return;
    }
    if (MODE == 3) {
        double* __restrict__ po = a.pout + row * a.n;
        if (s2) {
#pragma unroll
            for (int c = 0; c < 16; ++c) po[idx0 + N1 * 11 * c] = acc[c];
        }
        return;
    }
    if (MODE == 5) {
        // Per output residue k1 = k mod 217 (176 outputs, eleven lanes): the maximum, its first index, and the maximum of
        // the OTHER 175.  The second-peak search excludes fewer than 217 consecutive indices around the peak - at most one
        // per residue - so the row's maximum over the allowed indices is max_k1 (index allowed ? first : second), exactly,
        // and the winning row need not be transformed again (round 5; acq_rowtop2_peak_kernel).
        double b1 = -1.0, b2 = -1.0;
        int i1 = 0x7FFFFFFF;
        if (s2) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {   // (ascending index: the first of equal maxima stays, its twin is the second)
                const double v = acc[c];
                const bool gt = v > b1;
                b2 = fmax(b2, fmin(b1, v));
                b1 = gt ? v : b1;
                i1 = gt ? idx0 + N1 * 11 * c : i1;
            }
        }
        __syncthreads();
        double* s_b1 = reinterpret_cast<double*>(buf);
        double* s_b2 = s_b1 + CB * 11;
        int* s_i1 = reinterpret_cast<int*>(s_b2 + CB * 11);
        if (s2) {
            s_b1[tid] = b1;
            s_b2[tid] = b2;
            s_i1[tid] = i1;
        }
        __syncthreads();
        if (tid < CB) {
            double B1 = s_b1[tid * 11], B2 = s_b2[tid * 11];
            int I1 = s_i1[tid * 11];
#pragma unroll
            for (int d = 1; d < 11; ++d) {
                const double c1 = s_b1[tid * 11 + d], c2 = s_b2[tid * 11 + d];
                const int ci = s_i1[tid * 11 + d];
                const bool take = c1 > B1 || (c1 == B1 && ci < I1);
                B2 = fmax(fmax(B2, c2), take ? B1 : c1);
                B1 = take ? c1 : B1;
                I1 = take ? ci : I1;
            }
            const long long o = row * N1 + k10 + tid;
            a.t2_b1[o] = B1;
            a.t2_b2[o] = B2;
            a.t2_i1[o] = I1;
        }
        return;
    }
    double best = -1.0;
    int arg = 0x7FFFFFFF;
    if (s2) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {   // (ascending index: the first of equal maxima stays)
            if (acc[c] > best) {
                best = acc[c];
                arg = idx0 + N1 * 11 * c;
            }
        }
    }
    __syncthreads();
    double* s_v = reinterpret_cast<double*>(buf);
    int* s_i = reinterpret_cast<int*>(s_v + 2);
    wg_argmax<128>(best, arg, s_v, s_i, tid);
    if (tid == 0) {
        a.pmax[row * gridDim.x + blockIdx.x] = best;
        a.parg[row * gridDim.x + blockIdx.x] = arg;
    }
}

// ---- the plan the acquisition uses: n = 38192 = 176 x 217 (16*11, 7*31); other lengths keep the pass-per-radix path ----
#define F4_N1 F4W_N1       // columns: 7 x 31
#define F4_N2 F4W_N2       // rows: 16 x 11
#define F4_C 8             // columns kernel: 128 contiguous bytes per tile row; 27 KB of LDS, 166 VGPRs: five workgroups per CU
#define F4_CB 7            // rows kernel: 217 = 31 x 7 rows; 21 KB of LDS: seven workgroups per CU

bool sgx_fft4_supported(int64_t n) { return n == (int64_t)F4_N1 * F4_N2; }

static cplx* g_f4_sub[SGX_MAX_DEVICES][2] = {{nullptr, nullptr}};   // W_N1^t and W_N2^t per device
static int f4_ensure_sub_tables(int dev) {
    std::lock_guard<std::mutex> hold(g_wr_lock);
    if (g_f4_sub[dev][0]) return SGX_OK;
    const long double twopi = 2.0L * 3.14159265358979323846264338327950288L;
    const int len[2] = {F4_N1, F4_N2};
    for (int i = 0; i < 2; ++i) {
        std::vector<cplx> w((size_t)len[i]);
        for (int t = 0; t < len[i]; ++t) {
            const long double ang = -twopi * (long double)t / (long double)len[i];
            w[(size_t)t] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
        SGX_HIP(hipMalloc((void**)&g_f4_sub[dev][i], sizeof(cplx) * w.size()));
        SGX_HIP(hipMemcpy(g_f4_sub[dev][i], w.data(), sizeof(cplx) * w.size(), hipMemcpyHostToDevice));
    }
    return SGX_OK;
}
int sgx_fft4_row_blocks(void) { return F4_N1 / F4_CB; }
int sgx_fft4_residues(void) { return F4_N1; }

template <int MODE>
static void f4_launch_cols(const F4Args& a, int64_t rows, hipStream_t st) {
    dim3 grid(F4_N2 / F4_C, (unsigned)rows);
    if constexpr (MODE == 1) {
        static const bool plain = getenv("SGX_ACQ_XCD") && atoi(getenv("SGX_ACQ_XCD")) == 0;
        F4Args b = a;
        b.xcd_order = (!plain && !a.row_map && a.n_bins >= 1 && a.rows_per_prn == a.n_bins * a.n_blocks &&
                       rows % a.rows_per_prn == 0 && (grid.x * (unsigned)rows) % 8u == 0u) ? 1 : 0;
        fft4_cols217_kernel<1><<<grid, 128, 0, st>>>(b);
        return;
    }
    fft4_cols217_kernel<MODE><<<grid, 128, 0, st>>>(a);
}

template <int MODE, bool NB1>
static void f4_launch_rows_nb(const F4Args& a, int64_t rows, hipStream_t st) {
    dim3 grid(F4_N1 / F4_CB, (unsigned)rows);
    fft4_rows176_kernel<MODE, NB1><<<grid, 128, 0, st>>>(a);
}

template <int MODE>
static void f4_launch_rows(const F4Args& a, int64_t rows, hipStream_t st) {
    if (MODE == 0 || a.sum_blocks <= 1) f4_launch_rows_nb<MODE, true>(a, rows, st);
    else f4_launch_rows_nb<MODE, false>(a, rows, st);
}

// Forward transform of `rows` rows through the four-step kernels.  `work` holds the intermediate; the result (natural
// order) lands in `out`.  fuse (optional): product on load / reductions instead of a stored result, as FftFuse.
int sgx_fft4_forward(const FftPlan* p, const cplx* in, cplx* work, cplx* out, int64_t rows, hipStream_t st,
                     const Fft4Fuse* fuse) {
    if (!p->tw_hi || !sgx_fft4_supported(p->n) || rows < 1 || rows > 65535) {
        sgx_set_error("sgx_fft4_forward: length %lld / %lld rows not supported", (long long)p->n, (long long)rows);
        return SGX_E_ARG;
    }
    int dev = 0;
    SGX_HIP(hipGetDevice(&dev));
    for (int r : {16, 11, 7, 31}) {
        const int rc = ensure_roots(r);
        if (rc != SGX_OK) return rc;
    }
    if (dev < 0 || dev >= SGX_MAX_DEVICES) return SGX_E_ARG;
    {
        const int rc = f4_ensure_sub_tables(dev);
        if (rc != SGX_OK) return rc;
    }
    F4Args a;
    memset(&a, 0, sizeof(a));
    a.tw_sub = g_f4_sub[dev][0];
    a.tw_hi = p->tw_hi;
    a.tw_lo = p->tw_lo;
    a.lo_bits = p->lo_bits;
    a.n = p->n;
    a.nonzero_len = p->n;
    a.in = in;
    a.out = work;
    a.wr[0] = g_wr[dev][7];
    a.wr[1] = g_wr[dev][31];
    const int sum_blocks = (fuse && fuse->sum_blocks > 1) ? fuse->sum_blocks : 1;
    if (fuse && fuse->mul_x) {
        a.mul_x = fuse->mul_x;
        a.mul_f = fuse->mul_f;
        a.bin_map = fuse->bin_map;
        a.row_map = fuse->row_map;
        a.n_bins = fuse->n_bins;
        a.n_phi = fuse->n_phi;
        a.rows_per_prn = fuse->rows_per_prn;
        a.prn_base = fuse->prn_base;
        a.n_blocks = fuse->n_blocks;
        a.blocks_fast = fuse->blocks_fast;
        f4_launch_cols<1>(a, rows, st);
    } else {
        f4_launch_cols<0>(a, rows, st);
    }
    a.in = work;
    a.out = out;
    a.tw_sub = g_f4_sub[dev][1];
    a.wr[0] = g_wr[dev][16];
    a.wr[1] = g_wr[dev][11];
    if (fuse && fuse->second_out) {
        a.sec = fuse->sec;
        a.second_out = fuse->second_out;
        a.inv_n = fuse->inv_n;
        a.sum_blocks = sum_blocks;
        f4_launch_rows<4>(a, rows / sum_blocks, st);
    } else if (fuse && fuse->t2_b1) {
        a.t2_b1 = fuse->t2_b1;
        a.t2_b2 = fuse->t2_b2;
        a.t2_i1 = fuse->t2_i1;
        a.inv_n = fuse->inv_n;
        a.sum_blocks = sum_blocks;
        f4_launch_rows<5>(a, rows / sum_blocks, st);
    } else if (fuse && fuse->pmax) {
        a.pmax = fuse->pmax;
        a.parg = fuse->parg;
        a.inv_n = fuse->inv_n;
        a.sum_blocks = sum_blocks;
        f4_launch_rows<2>(a, rows / sum_blocks, st);
    } else if (fuse && fuse->pout) {
        a.pout = fuse->pout;
        a.inv_n = fuse->inv_n;
        a.sum_blocks = sum_blocks;
        f4_launch_rows<3>(a, rows / sum_blocks, st);
    } else {
        f4_launch_rows<0>(a, rows, st);
    }
    SGX_HIP(hipGetLastError());
    return SGX_OK;
}


// =====================================================================================================================
// Fine-frequency search on a two-kernel 2^22-point transform (acquisition.py:170-191): M = 1024 x 4096.
//   columns kernel   builds its input on the fly - two detected PRNs per complex row, (x - mean) * code(floor(ts k / tc))
//                    of the first in the real part, of the second in the imaginary part (acquisition.py:172-177), zeros
//                    beyond 10 ms - for 8 adjacent columns, runs their 1024-point transforms in LDS (radix 16, 16, 4),
//                    multiplies by W_M^(n2 k1) and stores element (k1, n2);
//   rows kernel      takes rows k1 and 1024 - k1 together (row 0 and row 512 alone), runs their 4096-point transforms in
//                    LDS (radix 16 x 3) and, without storing the spectrum, separates the two real signals
//                    X_a[k] = (Z[k] + conj(Z[M-k]))/2, X_b[k] = (Z[k] - conj(Z[M-k]))/(2i) - Z[M-k] of row k1 lives in
//                    row 1024 - k1 - and reduces |X|^2 over k in [4, M/2 - 4) to one (max, first index) per workgroup
//                    and detection (acquisition.py:182-187).  The spectrum crosses HBM once each way.
#define FF_N1 1024
#define FF_N2 4096
#define FF_C 8
#define FF_TPB 512

struct FineArgs {
    SgxSig x;
    const int8_t* codes;      // [32][1023]
    int det_prn[32];          // by value: no host-to-device copy between the coarse and the fine search
    int det_phase[32];
    int n_det;
    const int* det;           // or, device-led (round 4): the list in device memory, written by the coarse search's publish
                              // kernel - [0] = n_det, [1 + d] = PRN index, [33 + d] = code phase - so that the fine
                              // kernels are queued right behind the coarse ones without the host looking in between
    long long len;            // 10 N samples of signal, zeros beyond
    const long long* d_sum;   // sum of the record window (an integer for int8 samples, the bits of a double otherwise);
                              // mean = sum / n_mean (acquisition.py:59)
    double n_mean, ts, tc1;
    cplx* work;               // [rows][M] intermediate
    const cplx* tw_hi;        // two-level table of W_M
    const cplx* tw_lo;
    int lo_bits;
    const cplx* wr16;
    const cplx* wr4;
    const cplx* tw_n1;        // W_1024^t
    const cplx* tw_n2_hi;     // W_4096^(64 h), h < 64
    const cplx* tw_n2_lo;     // W_4096^l, l < 64
    long long lo, hi;         // arg-max range [lo, hi)
    double* pv;               // [n_det][FF_N1 / 2] per-workgroup maxima
    long long* pi;
    // device-led: the workgroup of fine_rows_kernel that finishes LAST folds the partial maxima into out_bi[d] and then
    // stores `seq` into *out_seq (both in the pinned result page the host spins on); det[FINE_DONE_SLOT] counts arrivals
    long long* out_bi;
    unsigned long long* out_seq;
    unsigned long long seq;
    // ... after copying stage_words dwords of the coarse search's outcome from device memory (stage_src, filled by the
    // publish kernel) to the same page (stage_dst): no kernel in front of the fine search writes host memory
    const int* stage_src;
    int* stage_dst;
    int stage_words;
};

#define FINE_DONE_SLOT 80

#define FA_NDET(a) ((a).det ? (a).det[0] : (a).n_det)
#define FA_PRN(a, d) ((a).det ? (a).det[1 + (d)] : (a).det_prn[d])
#define FA_PHASE(a, d) ((a).det ? (a).det[33 + (d)] : (a).det_phase[d])

__global__ __launch_bounds__(FF_TPB) void fine_cols_kernel(FineArgs a, int n_tiles_host) {
    const int n_det_k = FA_NDET(a);
    const int n_tiles = a.det ? (FF_N2 / FF_C) * ((n_det_k + 1) / 2) : n_tiles_host;
    if ((int)blockIdx.x >= n_tiles) return;
    extern __shared__ __attribute__((aligned(16))) char f4_smem[];
    cplx* __restrict__ buf = reinterpret_cast<cplx*>(f4_smem);   // [FF_N1][FF_C]
    cplx* __restrict__ twl = buf + FF_N1 * FF_C;                 // [FF_N1]
    const int tid = threadIdx.x;
    static_assert(FF_N1 == 2 * FF_TPB, "two staging registers");
    twl[tid] = a.tw_n1[tid];
    twl[tid + FF_TPB] = a.tw_n1[tid + FF_TPB];
    const double mean = (a.x.f64 ? __longlong_as_double(a.d_sum[0]) : (double)a.d_sum[0]) / a.n_mean;
    // Only the first len / FF_N2 (< 94 of 1024) rows of a column are non-zero.  `live` = elements of a tile that may be.
    constexpr int TILES_PER_ROW = FF_N2 / FF_C;
    const int live = (int)((a.len + FF_N2 - 1) / FF_N2) * FF_C;
    const bool few = live <= 2 * FF_TPB;      // the usual case: two elements per thread, the rest of the tile is zero
    // A workgroup holds one tile (128 KB of LDS: one workgroup per CU) and walks tiles t = blockIdx.x, + gridDim.x, ...
    // (spectrum t / 512, columns 8 (t % 512) ...).  The NEXT tile's two samples and code chips per thread - a division,
    // two dependent loads - are fetched during this tile's passes, and this tile's stores drain during the next one's.
    // (named registers: an array that lives across the loop is left in scratch memory by the compiler)
    double xa0 = 0.0, xa1 = 0.0, xb0 = 0.0, xb1 = 0.0, sa0 = 0.0, sa1 = 0.0, sb0 = 0.0, sb1 = 0.0;
    bool in0 = false, in1 = false;
#define FC_ONE(e_, c0_, pa_, pb_, ca_, cb_, two_, in_, xa_, xb_, sa_, sb_)                                    \
    do {                                                                                                      \
        const long long i_ = (long long)((e_) / FF_C) * FF_N2 + (c0_) + (e_) % FF_C;                          \
        in_ = i_ < a.len;                                                                                     \
        xa_ = xb_ = sa_ = sb_ = 0.0;                                                                          \
        if (in_) {                                                                                            \
            const double v_ = floor((a.ts * (double)(i_ + 1)) / a.tc1);       /* acquisition.py:172 (A9) */   \
            const int chip_ = (int)((long long)v_ % 1023);                                                    \
            xa_ = a.x.at((pa_) + i_);                                                                         \
            sa_ = (double)(ca_)[chip_];                                                                       \
            if (two_) {                                                                                       \
                xb_ = a.x.at((pb_) + i_);                                                                     \
                sb_ = (double)(cb_)[chip_];                                                                   \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#define FC_REQUEST(t_)                                                                                        \
    do {                                                                                                      \
        const int r_ = (t_) / TILES_PER_ROW, c0_ = ((t_) % TILES_PER_ROW) * FF_C;                              \
        const int d0_ = 2 * r_, d1_ = 2 * r_ + 1;                                                             \
        const bool two_ = d1_ < n_det_k;                                                                      \
        const long long pa_ = FA_PHASE(a, d0_), pb_ = two_ ? FA_PHASE(a, d1_) : pa_;                          \
        const int8_t* __restrict__ ca_ = a.codes + FA_PRN(a, d0_) * 1023;                                    \
        const int8_t* __restrict__ cb_ = two_ ? a.codes + FA_PRN(a, d1_) * 1023 : ca_;                        \
        FC_ONE(tid, c0_, pa_, pb_, ca_, cb_, two_, in0, xa0, xb0, sa0, sb0);                                  \
        FC_ONE(tid + FF_TPB, c0_, pa_, pb_, ca_, cb_, two_, in1, xa1, xb1, sa1, sb1);                         \
    } while (0)
    int t = blockIdx.x;
    if (few && t < n_tiles) FC_REQUEST(t);
    for (; t < n_tiles; t += gridDim.x) {
        const int r = t / TILES_PER_ROW, c0 = (t % TILES_PER_ROW) * FF_C;
        const bool two = 2 * r + 1 < n_det_k;
        if (few) {
            buf[tid] = in0 ? make_double2((xa0 - mean) * sa0, two ? (xb0 - mean) * sb0 : 0.0) : make_double2(0.0, 0.0);
            buf[tid + FF_TPB] = in1 ? make_double2((xa1 - mean) * sa1, two ? (xb1 - mean) * sb1 : 0.0) : make_double2(0.0, 0.0);
            if (t + (int)gridDim.x < n_tiles) FC_REQUEST(t + (int)gridDim.x);
            lds_sync();
            // first pass (radix 16, inputs n1 = j + 64 q): only q = 0, 1 can be non-zero (n1 < 128), so a butterfly is
            // out[q'] = v0 + v1 W_16^q' - no zero fill, no reads of zeros, no 16-point transform
            static_assert(FF_N1 / 16 * FF_C == FF_TPB, "one butterfly per thread");
            const int c = tid % FF_C, j = tid / FF_C;
            const cplx v0 = buf[j * FF_C + c], v1 = buf[(j + FF_N1 / 16) * FF_C + c];
            lds_sync();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cplx wq = a.wr16[q];
                buf[(j * 16 + q) * FF_C + c] = (q == 0) ? cadd(v0, v1) : cadd(v0, cmul(v1, wq));
            }
            lds_sync();
        } else {
            const int d0 = 2 * r, d1 = 2 * r + 1;
            const long long pa = FA_PHASE(a, d0), pb = two ? FA_PHASE(a, d1) : pa;
            const int8_t* __restrict__ ca = a.codes + FA_PRN(a, d0) * 1023;
            const int8_t* __restrict__ cb = two ? a.codes + FA_PRN(a, d1) * 1023 : ca;
            for (int e = tid; e < FF_N1 * FF_C; e += FF_TPB) {
                bool in_;
                double xa_, xb_, sa_, sb_;
                FC_ONE(e, c0, pa, pb, ca, cb, two, in_, xa_, xb_, sa_, sb_);
                buf[e] = in_ ? make_double2((xa_ - mean) * sa_, two ? (xb_ - mean) * sb_ : 0.0) : make_double2(0.0, 0.0);
            }
            lds_sync();
            const TwDirect tw1{twl};
            lds_radix_pass<FF_N1, 16, 1, FF_C, FF_C, 1, FF_TPB, true>(buf, tw1, a.wr16, tid);
        }
        const TwDirect tw{twl};
        lds_radix_pass<FF_N1, 16, 16, FF_C, FF_C, 1, FF_TPB, true>(buf, tw, a.wr16, tid);
        lds_radix_pass<FF_N1, 4, 256, FF_C, FF_C, 1, FF_TPB, true>(buf, tw, a.wr4, tid);
        // element (k1, n2) times W_M^(n2 k1): a thread keeps its column and advances k1 by FF_TPB / FF_C; table look-ups
        // for its first element and for the step factor, the other 15 by recurrence (a rolled loop of look-ups runs one
        // memory round trip per element)
        cplx* __restrict__ out = a.work + (long long)r * FF_N1 * FF_N2;
        static_assert(FF_TPB % FF_C == 0 && FF_N1 % (FF_TPB / FF_C) == 0, "a thread keeps its column");
        constexpr int STEP = FF_TPB / FF_C;
        const int n2 = c0 + tid % FF_C, k1b = tid / FF_C;
        auto look = [&](long long tt) { return cmul(a.tw_hi[tt >> a.lo_bits], a.tw_lo[tt & ((1ll << a.lo_bits) - 1)]); };
        cplx w = look((long long)n2 * k1b);
        const cplx ws = look((long long)n2 * STEP);
#pragma unroll
        for (int i = 0; i < FF_N1 / STEP; ++i) {
            const int k1 = k1b + i * STEP;
            out[(long long)k1 * FF_N2 + n2] = cmul(buf[k1 * FF_C + tid % FF_C], w);
            w = cmul(w, ws);
        }
        lds_sync();   // (the next tile's samples go where this one's spectrum is being read)
    }
#undef FC_REQUEST
#undef FC_ONE
}

__global__ __launch_bounds__(FF_TPB) void fine_rows_kernel(FineArgs a, int n_pairs_host) {
    const int n_det_k = FA_NDET(a);
    const int n_pairs = a.det ? (FF_N1 / 2) * ((n_det_k + 1) / 2) : n_pairs_host;
    if ((int)blockIdx.x >= n_pairs) {
        if (a.out_seq && n_pairs == 0 && blockIdx.x == 0) {   // nothing detected: the coarse outcome and the word
            for (int i = threadIdx.x; i < a.stage_words; i += FF_TPB) a.stage_dst[i] = a.stage_src[i];
            __threadfence_system();   // every wave's page stores are out before the barrier: the word's release covers thread 0's only
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(a.out_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char f4_smem[];
    constexpr int RP = FF_N2 + FF_N2 / 16;                       // a row with one spare element per 16 (lds_radix_pass)
    cplx* __restrict__ buf = reinterpret_cast<cplx*>(f4_smem);   // [2][RP]
    cplx* __restrict__ thi = buf + 2 * RP;                       // [64]
    cplx* __restrict__ tlo = thi + 64;                           // [64]
    double* __restrict__ s_v = reinterpret_cast<double*>(tlo + 64);                  // [2][FF_TPB / 64]
    long long* __restrict__ s_i = reinterpret_cast<long long*>(s_v + 2 * (FF_TPB / 64));
    const int tid = threadIdx.x;
    if (tid < 64) {
        thi[tid] = a.tw_n2_hi[tid];
        tlo[tid] = a.tw_n2_lo[tid];
    }
    // A workgroup holds two rows (128 KB of LDS: one workgroup per CU) and walks pairs p = blockIdx.x, + gridDim.x, ...:
    // the NEXT pair's rows are requested as soon as this pair's have been handed to LDS and arrive during its three
    // passes (one workgroup per pair loaded, transformed and reduced one after the other with nothing else on the CU).
    // Pair p: spectrum r = p / 512, rows bx and 1024 - bx, bx = p % 512; rows 0 and 512 are their OWN mirrors and share
    // bx = 0.
    constexpr int PER = FF_N2 / FF_TPB;
    static_assert(FF_N2 % FF_TPB == 0, "whole rounds");
    constexpr int HALF = FF_N1 / 2;
    static_assert(PER == 8, "sixteen staging registers");
    // (named: an array that lives across the loop is left in scratch memory by the compiler)
    cplx a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
#define FR_REQUEST(p_)                                                                                       \
    do {                                                                                                     \
        const int r_ = (p_) / HALF, bx_ = (p_) % HALF;                                                       \
        const cplx* __restrict__ pa_ = a.work + ((long long)r_ * FF_N1 + bx_) * FF_N2 + tid;                 \
        const cplx* __restrict__ pb_ = a.work + ((long long)r_ * FF_N1 + (bx_ == 0 ? HALF : FF_N1 - bx_)) * FF_N2 + tid; \
        a0 = pa_[0 * FF_TPB]; a1 = pa_[1 * FF_TPB]; a2 = pa_[2 * FF_TPB]; a3 = pa_[3 * FF_TPB];               \
        a4 = pa_[4 * FF_TPB]; a5 = pa_[5 * FF_TPB]; a6 = pa_[6 * FF_TPB]; a7 = pa_[7 * FF_TPB];               \
        b0 = pb_[0 * FF_TPB]; b1 = pb_[1 * FF_TPB]; b2 = pb_[2 * FF_TPB]; b3 = pb_[3 * FF_TPB];               \
        b4 = pb_[4 * FF_TPB]; b5 = pb_[5 * FF_TPB]; b6 = pb_[6 * FF_TPB]; b7 = pb_[7 * FF_TPB];               \
    } while (0)
    int p = blockIdx.x;
    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = b0 = b1 = b2 = b3 = b4 = b5 = b6 = b7 = make_double2(0.0, 0.0);
    if (p < n_pairs) FR_REQUEST(p);
    for (; p < n_pairs; p += gridDim.x) {
        const int r = p / HALF, bx = p % HALF;
        const bool single = (bx == 0);
        const int rowA = bx, rowB = single ? HALF : FF_N1 - bx;
        {
            auto at = [](int n) { return n + (n >> 4); };
            cplx* __restrict__ A = buf + at(tid);          // (FF_TPB is a multiple of 16: round i is 17 FF_TPB / 16 further on)
            cplx* __restrict__ B = A + RP;
            constexpr int ST = FF_TPB + FF_TPB / 16;
            A[0 * ST] = a0; A[1 * ST] = a1; A[2 * ST] = a2; A[3 * ST] = a3; A[4 * ST] = a4; A[5 * ST] = a5; A[6 * ST] = a6; A[7 * ST] = a7;
            B[0 * ST] = b0; B[1 * ST] = b1; B[2 * ST] = b2; B[3 * ST] = b3; B[4 * ST] = b4; B[5 * ST] = b5; B[6 * ST] = b6; B[7 * ST] = b7;
        }
        if (p + (int)gridDim.x < n_pairs) FR_REQUEST(p + (int)gridDim.x);
        lds_sync();
        const TwTwoLevel tw{thi, tlo};
        lds_radix_pass<FF_N2, 16, 1, 2, 1, RP, FF_TPB, false, 4>(buf, tw, a.wr16, tid);
        lds_radix_pass<FF_N2, 16, 16, 2, 1, RP, FF_TPB, false, 4>(buf, tw, a.wr16, tid);
        lds_radix_pass<FF_N2, 16, 256, 2, 1, RP, FF_TPB, false, 4>(buf, tw, a.wr16, tid);
        // Z[k1 + 1024 k2] = row k1, element k2;  Z[M - k] = row (1024 - k1) mod 1024, element 4095 - k2 (k1 > 0) or
        // 4096 - k2 (k1 = 0, k2 > 0)
        double best[2] = {-1.0, -1.0};
        long long arg[2] = {a.lo, a.lo};
        for (int rr = 0; rr < 2; ++rr) {
            const int k1 = rr == 0 ? rowA : rowB;
            const cplx* __restrict__ me = buf + rr * RP;
            const cplx* __restrict__ other = buf + (single ? rr : (1 - rr)) * RP;
            for (int k2 = tid; k2 < FF_N2; k2 += FF_TPB) {
                const long long k = (long long)k1 + (long long)FF_N1 * k2;
                if (k < a.lo || k >= a.hi) continue;
                const int ko = (k1 == 0) ? FF_N2 - k2 : FF_N2 - 1 - k2;     // k >= 4 excludes k1 = 0, k2 = 0
                const cplx z = me[k2 + (k2 >> 4)];
                const cplx w = other[ko + (ko >> 4)];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const double sgn = d ? -1.0 : 1.0;
                    const double re = z.x + sgn * w.x, im = z.y - sgn * w.y;   // z +- conj(w)
                    const double v = re * re + im * im;
                    if (v > best[d] || (v == best[d] && k < arg[d])) {
                        best[d] = v;
                        arg[d] = k;
                    }
                }
            }
        }
        // (maximum, first index) of each detection: inside the waves by shuffles, across them by one LDS hop
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            double bv = best[d];
            long long bi = arg[d];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(bv, off);
                const long long oi = __shfl_down(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if ((tid & 63) == 0) {
                s_v[d * (FF_TPB / 64) + (tid >> 6)] = bv;
                s_i[d * (FF_TPB / 64) + (tid >> 6)] = bi;
            }
        }
        lds_sync();
        if (tid < 2) {
            const int det = 2 * r + tid;
            if (det < n_det_k) {
                double bv = s_v[tid * (FF_TPB / 64)];
                long long bi = s_i[tid * (FF_TPB / 64)];
                for (int w = 1; w < FF_TPB / 64; ++w) {
                    const double ov = s_v[tid * (FF_TPB / 64) + w];
                    const long long oi = s_i[tid * (FF_TPB / 64) + w];
                    if (ov > bv || (ov == bv && oi < bi)) {
                        bv = ov;
                        bi = oi;
                    }
                }
                if (a.out_seq) {   // device-led: read by the last workgroup of THIS launch - written through, no fence
                    __hip_atomic_store(a.pv + (long long)det * HALF + bx, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.pi + (long long)det * HALF + bx, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    a.pv[(long long)det * HALF + bx] = bv;
                    a.pi[(long long)det * HALF + bx] = bi;
                }
            }
        }
        // (the next pair's rows overwrite buf and the slots: everybody is past both by the barrier of the next turn...
        // which comes AFTER those stores; so one more here)
        lds_sync();
    }
    if (!a.out_seq) return;
    // device-led: the last workgroup to arrive finishes the search (the host's loop of rounds 2-3: per detection the
    // FIRST maximum over the workgroups' partials) and tells the host
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        const unsigned n_active = (unsigned)((int)gridDim.x < n_pairs ? (int)gridDim.x : n_pairs);
        unsigned* ctr = reinterpret_cast<unsigned*>(const_cast<int*>(a.det)) + FINE_DONE_SLOT;
        const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket + 1u == n_active;
        if (s_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = tid; i < a.stage_words; i += FF_TPB) a.stage_dst[i] = a.stage_src[i];
    __threadfence_system();   // (each wave drains its own page stores; the barriers below then order them before the word)
    __syncthreads();
    static_assert(FF_N1 / 2 == FF_TPB, "one partial per thread");
    for (int d = 0; d < n_det_k; ++d) {
        double bv = __hip_atomic_load(a.pv + (long long)d * (FF_N1 / 2) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long bi = __hip_atomic_load(a.pi + (long long)d * (FF_N1 / 2) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o >= 1; o >>= 1) {
            const double ov = __shfl_xor(bv, o);
            const long long oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            s_v[tid >> 6] = bv;
            s_i[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < FF_TPB / 64; ++w) {
                const double ov = s_v[w];
                const long long oi = s_i[w];
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            a.out_bi[d] = bi;
        }
        __syncthreads();
    }
    if (tid == 0) {
        __hip_atomic_store(a.out_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#undef FR_REQUEST

static cplx* g_ff_tab[SGX_MAX_DEVICES][3] = {{nullptr, nullptr, nullptr}};   // W_1024^t | W_4096^(64 h) | W_4096^l

bool sgx_fft_fine_supported(int64_t npts) { return npts == (int64_t)FF_N1 * FF_N2; }
int sgx_fft_fine_partials(void) { return FF_N1 / 2; }

// Fine search of n_det detections (two per complex row) on the 2^22-point two-kernel transform.  `plan` = the 2^22
// plan (its two-level table of W_M is used for the inter-step twiddles).  Fills pv / pi [n_det][sgx_fft_fine_partials()].
int sgx_fft_fine_search(const FftPlan* plan, SgxSig x, const int8_t* codes, const int* det_prn,
                        const int* det_phase, int n_det, long long len, const long long* d_sum, double n_mean, double ts,
                        double tc1, cplx* work, long long lo, long long hi, double* pv, long long* pi, hipStream_t st,
                        const int* d_det, long long* out_bi, unsigned long long* out_seq, unsigned long long seq,
                        const int* stage_src, int* stage_dst, int stage_words) {
    // d_det != nullptr: device-led - the detection list is in device memory (n_det here = the most it can hold)
    if (!plan->tw_hi || !sgx_fft_fine_supported(plan->n) || n_det < 1 || n_det > 32) {
        sgx_set_error("sgx_fft_fine_search: %lld points not supported", (long long)plan->n);
        return SGX_E_ARG;
    }
    int dev = 0;
    SGX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= SGX_MAX_DEVICES) return SGX_E_ARG;
    for (int r : {16, 4}) {
        const int rc = ensure_roots(r);
        if (rc != SGX_OK) return rc;
    }
    {
        std::lock_guard<std::mutex> hold(g_wr_lock);
        if (!g_ff_tab[dev][0]) {
            const long double twopi = 2.0L * 3.14159265358979323846264338327950288L;
            std::vector<cplx> t1(FF_N1), th(64), tl(64);
            for (int t = 0; t < FF_N1; ++t) {
                const long double ang = -twopi * (long double)t / (long double)FF_N1;
                t1[(size_t)t] = make_double2((double)cosl(ang), (double)sinl(ang));
            }
            for (int t = 0; t < 64; ++t) {
                const long double ah = -twopi * (long double)(64 * t) / (long double)FF_N2;
                const long double al = -twopi * (long double)t / (long double)FF_N2;
                th[(size_t)t] = make_double2((double)cosl(ah), (double)sinl(ah));
                tl[(size_t)t] = make_double2((double)cosl(al), (double)sinl(al));
            }
            SGX_HIP(hipMalloc((void**)&g_ff_tab[dev][0], sizeof(cplx) * FF_N1));
            SGX_HIP(hipMalloc((void**)&g_ff_tab[dev][1], sizeof(cplx) * 64));
            SGX_HIP(hipMalloc((void**)&g_ff_tab[dev][2], sizeof(cplx) * 64));
            SGX_HIP(hipMemcpy(g_ff_tab[dev][0], t1.data(), sizeof(cplx) * FF_N1, hipMemcpyHostToDevice));
            SGX_HIP(hipMemcpy(g_ff_tab[dev][1], th.data(), sizeof(cplx) * 64, hipMemcpyHostToDevice));
            SGX_HIP(hipMemcpy(g_ff_tab[dev][2], tl.data(), sizeof(cplx) * 64, hipMemcpyHostToDevice));
        }
    }
    FineArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.codes = codes;
    for (int d = 0; d < n_det && !d_det; ++d) {
        a.det_prn[d] = det_prn[d];
        a.det_phase[d] = det_phase[d];
    }
    a.n_det = n_det;
    a.det = d_det;
    a.len = len;
    a.d_sum = d_sum;
    a.n_mean = n_mean;
    a.ts = ts;
    a.tc1 = tc1;
    a.work = work;
    a.tw_hi = plan->tw_hi;
    a.tw_lo = plan->tw_lo;
    a.lo_bits = plan->lo_bits;
    a.wr16 = g_wr[dev][16];
    a.wr4 = g_wr[dev][4];
    a.tw_n1 = g_ff_tab[dev][0];
    a.tw_n2_hi = g_ff_tab[dev][1];
    a.tw_n2_lo = g_ff_tab[dev][2];
    a.lo = lo;
    a.hi = hi;
    a.pv = pv;
    a.pi = pi;
    a.out_bi = d_det ? out_bi : nullptr;
    a.out_seq = d_det ? out_seq : nullptr;
    a.stage_src = stage_src;
    a.stage_dst = stage_dst;
    a.stage_words = (d_det && stage_src && stage_dst) ? stage_words : 0;
    a.seq = seq;
    const int n_rows = (n_det + 1) / 2;
    const size_t lds_c = sizeof(cplx) * (FF_N1 * FF_C + FF_N1);
    const size_t lds_r = sizeof(cplx) * (2 * (FF_N2 + FF_N2 / 16) + 128) + (sizeof(double) + sizeof(long long)) * 2 * (FF_TPB / 64);
    static std::atomic<bool> once[SGX_MAX_DEVICES];
    if (!once[dev].load()) {
        hipFuncSetAttribute((const void*)fine_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        hipFuncSetAttribute((const void*)fine_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
        once[dev].store(true);
    }
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    {
        // persistent workgroups, one per CU (128 KB of LDS each), walking the (spectrum, column tile) list
        const int n_tiles = (FF_N2 / FF_C) * n_rows;
        fine_cols_kernel<<<n_tiles < cus ? n_tiles : cus, FF_TPB, lds_c, st>>>(a, n_tiles);
    }
    {
        // persistent workgroups, one per CU (128 KB of LDS each), walking the (spectrum, row pair) list
        const int n_pairs = (FF_N1 / 2) * n_rows;
        const int grid = n_pairs < cus ? n_pairs : cus;
        fine_rows_kernel<<<grid, FF_TPB, lds_r, st>>>(a, n_pairs);
    }
    SGX_HIP(hipGetLastError());
    return SGX_OK;
}
