// Synthetic IF generator on the device: bit-identical twin of softgnss-python_amd/synth.py.
// Integer arithmetic only. HBM-write bound: 16 samples per lane, one 16-byte store each.
#include "sgx_internal.h"

int sgx_if_alloc_internal(sgx_ctx* c, size_t n, sgx_if** out);

#define GOLDEN 0x9E3779B97F4A7C15ull

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

struct SynthArgs {   // the scene without its navigation tables (kernel arguments are limited to 4 KiB)
    uint64_t seed;
    int32_t n_sats;
    int32_t nav_mode;
    sgx_sat sats[SGX_MAX_SATS];
    int16_t cos_lut[256];
};

__global__ __launch_bounds__(256) void synth_kernel(int8_t* __restrict__ out, uint64_t offset, uint64_t n,
                                                    const int8_t* __restrict__ codes,
                                                    const uint8_t* __restrict__ nav_bits, SynthArgs a) {
    __shared__ int8_t s_code[SGX_MAX_SATS][1024];
    __shared__ int16_t s_lut[256];
    const int nsat = a.n_sats;
    for (int i = threadIdx.x; i < nsat * 1024; i += blockDim.x) {
        const int s = i >> 10, k = i & 1023;
        s_code[s][k] = (k < 1023) ? codes[(a.sats[s].prn - 1) * 1023 + k] : 0;
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = a.cos_lut[i];
    __syncthreads();

    const uint64_t groups = (n + 15) / 16;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint64_t idx = offset + g * 16 + b;
            const uint64_t h = splitmix64(a.seed + (idx + 1) * GOLDEN);
            const int s4 = (int)(h & 0xFF) + (int)((h >> 8) & 0xFF) + (int)((h >> 16) & 0xFF) +
                           (int)((h >> 24) & 0xFF);
            int acc = ((s4 - 510) * 35) >> 8;
            for (int s = 0; s < nsat; ++s) {
                const sgx_sat& st = a.sats[s];
                const uint64_t cp = idx * st.code_fcw + st.code_c0;
                const uint64_t chipw = cp >> 32;
                const uint32_t chip = (uint32_t)(chipw % 1023ull);
                const uint64_t bit = chipw / (1023ull * 20ull);
                int nav;
                if (a.nav_mode) {
                    const unsigned tb = (unsigned)(bit & 2047);
                    nav = 2 * (int)((nav_bits[s * 256 + (tb >> 3)] >> (tb & 7)) & 1) - 1;
                } else {
                    const uint64_t navh = splitmix64(st.nav_seed + (bit + 1) * GOLDEN);
                    nav = 1 - 2 * (int)(navh & 1);
                }
                const uint32_t ph = st.car_ph0 + (uint32_t)(idx * (uint64_t)st.car_fcw);
                const int cv = s_lut[ph >> 24];
                acc += (st.amp * (int)s_code[s][chip] * nav * cv + 64) >> 7;
            }
            acc = acc < -127 ? -127 : (acc > 127 ? 127 : acc);
            w[b >> 2] |= ((uint32_t)(acc & 0xFF)) << ((b & 3) * 8);
        }
        if (g * 16 + 16 <= n) {
            *reinterpret_cast<uint4*>(out + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; b < 16 && g * 16 + b < n; ++b)
                out[g * 16 + b] = (int8_t)((w[b >> 2] >> ((b & 3) * 8)) & 0xFF);
        }
    }
}

extern "C" int sgx_if_synth(sgx_ctx* c, const sgx_scene* scene, uint64_t offset, size_t n, sgx_if** out) {
    SGX_CHECK_ARG(c && scene && out);
    SGX_CHECK_ARG(scene->n_sats >= 0 && scene->n_sats <= SGX_MAX_SATS);
    for (int s = 0; s < scene->n_sats; ++s) SGX_CHECK_ARG(scene->sats[s].prn >= 1 && scene->sats[s].prn <= 32);
    SGX_HIP(hipSetDevice(c->device));
    sgx_if* r = nullptr;
    int rc = sgx_if_alloc_internal(c, n, &r);
    if (rc != SGX_OK) return rc;
    SynthArgs a;
    a.seed = scene->seed;
    a.n_sats = scene->n_sats;
    a.nav_mode = scene->nav_mode;
    memcpy(a.sats, scene->sats, sizeof(a.sats));
    memcpy(a.cos_lut, scene->cos_lut, sizeof(a.cos_lut));
    uint8_t* d_nav = (uint8_t*)c->d_small + 512 * 1024;   // upper half of the context's small device area
    SGX_HIP(hipMemcpyAsync(d_nav, scene->nav_bits, sizeof(scene->nav_bits), hipMemcpyHostToDevice, c->stream));
    const uint64_t groups = (n + 15) / 16;
    int blocks = (int)((groups + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipEventRecord(c->ev[0], c->stream);
    synth_kernel<<<blocks, 256, 0, c->stream>>>(r->d, offset, (uint64_t)n, c->d_codes, d_nav, a);
    hipEventRecord(c->ev[1], c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        sgx_if_free(c, r);
        sgx_set_error("synth kernel failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    hipEventElapsedTime(&c->timing.synth_ms, c->ev[0], c->ev[1]);
    *out = r;
    return SGX_OK;
}

// ---- measured HBM rates for the roofline report ("also report vs. a measured stream peak") --
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* __restrict__ src, size_t n16,
                                                          unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) *sink = acc;   // keeps the loads alive; practically never true
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                          size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int sgx_stream_rates(sgx_ctx* c, size_t bytes, int reps, double* read_gbs, double* copy_gbs) {
    SGX_CHECK_ARG(c && read_gbs && copy_gbs && bytes >= (1u << 20) && reps >= 1);
    SGX_HIP(hipSetDevice(c->device));
    const size_t n16 = bytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    SGX_HIP(hipMalloc((void**)&a, n16 * 16));
    if (hipMalloc((void**)&b, n16 * 16) != hipSuccess) {
        hipFree(a);
        sgx_set_error("hipMalloc of %zu bytes failed in sgx_stream_rates", n16 * 16);
        return SGX_E_NOMEM;
    }
    hipStream_t st = c->stream;
    hipMemsetAsync(a, 0x5A, n16 * 16, st);
    hipMemsetAsync(b, 0, n16 * 16, st);
    const int blocks = 256 * 16;
    float ms_r = 0.f, ms_c = 0.f;
    stream_read_kernel<<<blocks, 256, 0, st>>>(a, n16, (unsigned*)b);
    hipEventRecord(c->ev[0], st);
    for (int i = 0; i < reps; ++i) stream_read_kernel<<<blocks, 256, 0, st>>>(a, n16, (unsigned*)b);
    hipEventRecord(c->ev[1], st);
    stream_copy_kernel<<<blocks, 256, 0, st>>>(a, b, n16);
    hipEventRecord(c->ev[2], st);
    for (int i = 0; i < reps; ++i) stream_copy_kernel<<<blocks, 256, 0, st>>>(a, b, n16);
    hipEventRecord(c->ev[3], st);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) {
        hipEventElapsedTime(&ms_r, c->ev[0], c->ev[1]);
        hipEventElapsedTime(&ms_c, c->ev[2], c->ev[3]);
    }
    hipFree(a);
    hipFree(b);
    if (e != hipSuccess) {
        sgx_set_error("stream rate kernels failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    *read_gbs = (double)(n16 * 16) * reps / (ms_r * 1e-3) / 1e9;
    *copy_gbs = 2.0 * (double)(n16 * 16) * reps / (ms_c * 1e-3) / 1e9;   // bytes read + bytes written
    return SGX_OK;
}
