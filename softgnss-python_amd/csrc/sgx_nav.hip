// Bit synchronisation + preamble search on the tracking output (reference postNavigation.py:443-631,
// SURVEY.md section 8(f) item 1): the step that consumes I_P right after TrackingResult.track.
//
//   device  c[ch][t] = sum_{k<160} sign(I_P[ch][t+k]) * preamble_ms[k]   (the reference correlates against a
//           preamble zero-padded to the full record length, O(L^2); only 160 taps are non-zero)
//   host    candidates |c| > 153 in increasing order, a partner exactly 6000 ms later, parity of the TLM and
//           HOW words on 20-ms sums (navPartyChk) - a few dozen candidates per channel.
#include <math.h>

#include "sgx_internal.h"

#define NAV_TAPS 160

__global__ __launch_bounds__(256) void nav_corr_kernel(const double* __restrict__ ip, short* __restrict__ corr,
                                                       int ms, int start) {
    __shared__ signed char s_b[256 + NAV_TAPS];
    const int ch = blockIdx.y;
    const int len = ms - start;
    const int t0 = blockIdx.x * 256;
    const double* __restrict__ row = ip + (long long)ch * ms + start;
    for (int i = threadIdx.x; i < 256 + NAV_TAPS; i += 256) {
        const int t = t0 + i;
        s_b[i] = (t < len) ? (row[t] > 0.0 ? 1 : -1) : 0;   // bits > 0 -> 1, <= 0 -> -1; past the end: 0
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= len) return;
    const int pre[8] = {1, -1, -1, -1, 1, -1, 1, 1};   // postNavigation.py:552
    int acc = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 20; ++k) sum += s_b[threadIdx.x + 20 * b + k];
        acc += pre[b] * sum;
    }
    corr[(long long)ch * ms + start + t] = (short)acc;
}

// candidate selection and parity checks on the host: csrc/sgx_navhost.cpp
int sgx_nav_select(const double* I_P, const short* corr, int32_t n_ch, int32_t ms, int32_t search_start,
                   int32_t* firstSubFrame);

extern "C" int sgx_find_preambles(sgx_ctx* c, const double* I_P, int32_t n_ch, int32_t ms, int32_t search_start,
                                  int32_t* firstSubFrame) {
    SGX_CHECK_ARG(c && I_P && firstSubFrame && n_ch >= 1 && ms >= 1 && search_start >= 0 && search_start < ms);
    SGX_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t n = (size_t)n_ch * (size_t)ms;
    double* d_ip = nullptr;
    short* d_c = nullptr;
    SGX_HIP(hipMalloc((void**)&d_ip, n * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_c, n * sizeof(short));
    if (e != hipSuccess) {
        hipFree(d_ip);
        sgx_set_error("hipMalloc failed in sgx_find_preambles");
        return SGX_E_NOMEM;
    }
    std::vector<short> corr(n);
    e = hipMemcpyAsync(d_ip, I_P, n * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        dim3 grid((unsigned)((ms - search_start + 255) / 256), (unsigned)n_ch);
        nav_corr_kernel<<<grid, 256, 0, st>>>(d_ip, d_c, ms, search_start);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(corr.data(), d_c, n * sizeof(short), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_ip);
    hipFree(d_c);
    if (e != hipSuccess) {
        sgx_set_error("preamble correlation failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    return sgx_nav_select(I_P, corr.data(), n_ch, ms, search_start, firstSubFrame);
}
