// Host side of libsgx.so: error text, exact host helpers, device context, IF records, RCCL gather.
// Compiled with -ffp-contract=off: the index math below must round exactly like the reference's
// numpy expressions.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <chrono>
#include <string>

#include "sgx_internal.h"
#include "sgx_trk_math.h"

static thread_local char g_err[512] = "";

void sgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sgx_version(void) { return SGX_VERSION_STR; }

extern "C" int sgx_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return SGX_E_ARG;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return SGX_OK;
}

// ---- exact host helpers ---------------------------------------------------------------------

// G2 delays of PRN 1..32 (reference initialize.py:251-254 keeps 51 entries; only 32 reachable).
static const int kG2Delay[32] = {5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252,
                                 254, 255, 256, 257, 258, 469, 470, 471, 472, 473, 474,
                                 509, 512, 513, 514, 515, 516, 859, 860, 861, 862};

// Gold code of PRN index prn0 as +-1 chips. Bit-level statement of initialize.py:234-302:
// registers start all-ones, output = stage 10, G1 feedback 3^10, G2 feedback 2^3^6^8^9^10,
// G2 delayed by kG2Delay, chip = +1 where g1^g2 == 1.
int sgx_host_ca_code(int prn0, int8_t* out) {
    if (prn0 < 0 || prn0 > 31) return SGX_E_ARG;
    uint32_t r1 = 0x3FF, r2 = 0x3FF;   // bit i = stage i+1
    int8_t g1[1023], g2[1023];
    for (int i = 0; i < 1023; ++i) {
        g1[i] = (r1 >> 9) & 1;
        g2[i] = (r2 >> 9) & 1;
        uint32_t f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1;
        uint32_t f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^ (r2 >> 8) ^ (r2 >> 9)) & 1;
        r1 = ((r1 << 1) | f1) & 0x3FF;
        r2 = ((r2 << 1) | f2) & 0x3FF;
    }
    const int d = kG2Delay[prn0];
    for (int i = 0; i < 1023; ++i) {
        int j = i - d;
        if (j < 0) j += 1023;
        out[i] = (g1[i] ^ g2[j]) ? 1 : -1;
    }
    return SGX_OK;
}

int64_t sgx_host_samples_per_code(const sgx_settings* s) {
    // initialize.py:185: long(round(fs / (fc / codeLength))), numpy round = half to even
    return (int64_t)nearbyint(s->samplingFreq / (s->codeFreqBasis / (double)s->codeLength));
}

extern "C" int sgx_samples_per_code(const sgx_settings* s, int64_t* n) {
    SGX_CHECK_ARG(s && n);
    *n = sgx_host_samples_per_code(s);
    return SGX_OK;
}

extern "C" int sgx_generate_ca_code(int32_t prn0, double* out) {
    SGX_CHECK_ARG(out);
    int8_t c[1023];
    if (sgx_host_ca_code(prn0, c) != SGX_OK) {
        sgx_set_error("prn index %d outside 0..31", prn0);   // reference asserts (initialize.py:250)
        return SGX_E_ARG;
    }
    for (int i = 0; i < 1023; ++i) out[i] = (double)c[i];
    return SGX_OK;
}

extern "C" int sgx_make_ca_table(const sgx_settings* s, double* out) {
    SGX_CHECK_ARG(s && out);
    const int64_t n = sgx_host_samples_per_code(s);
    SGX_CHECK_ARG(n > 0 && s->codeLength == 1023);
    const double ts = 1.0 / s->samplingFreq;
    const double tc = 1.0 / s->codeFreqBasis;
    std::vector<int> idx((size_t)n);
    for (int64_t k = 1; k <= n; ++k) {
        const double v = (ts * (double)k) / tc;   // initialize.py:222: multiply, then divide
        idx[(size_t)(k - 1)] = (int)ceil(v) - 1;
    }
    idx[(size_t)(n - 1)] = 1022;                  // initialize.py:226
    for (int p = 0; p < 32; ++p) {
        int8_t c[1023];
        sgx_host_ca_code(p, c);
        double* row = out + (size_t)p * (size_t)n;
        for (int64_t k = 0; k < n; ++k) {
            const int j = idx[(size_t)k];
            if (j < 0 || j > 1022) {
                sgx_set_error("code index %d out of range at sample %lld", j, (long long)k);
                return SGX_E_ARG;
            }
            row[k] = (double)c[j];
        }
    }
    return SGX_OK;
}

extern "C" int sgx_calc_loop_coef(double lbw, double zeta, double k, double* tau1, double* tau2) {
    SGX_CHECK_ARG(tau1 && tau2);
    const double wn = lbw * 8.0 * zeta / (4.0 * (zeta * zeta) + 1);   // initialize.py:321
    *tau1 = k / (wn * wn);
    *tau2 = 2.0 * zeta / wn;
    return SGX_OK;
}

// ---- device context --------------------------------------------------------------------------

extern "C" int sgx_device_count(int* n) {
    SGX_CHECK_ARG(n);
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        sgx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    *n = c;
    return SGX_OK;
}

extern "C" int sgx_ctx_create(const sgx_settings* s, int device, sgx_ctx** out) {
    return sgx_ctx_create_prio(s, device, 0, out);
}

static int ctx_build(sgx_ctx* c, int priority) {
    if (priority == 0) {
        SGX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    } else {
        int least = 0, greatest = 0;   // numerically: greatest priority <= least priority
        SGX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        SGX_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority < 0 ? greatest : least));
    }
    for (int i = 0; i < 6; ++i) SGX_HIP(hipEventCreate(&c->ev[i]));
    std::vector<int8_t> codes(32 * 1023);
    for (int p = 0; p < 32; ++p) sgx_host_ca_code(p, codes.data() + p * 1023);
    SGX_HIP(hipMalloc((void**)&c->d_codes, codes.size()));
    SGX_HIP(hipMemcpy(c->d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
    SGX_HIP(hipMalloc(&c->d_small, 1 << 20));
    SGX_HIP(hipHostMalloc(&c->h_small, 1 << 20, hipHostMallocDefault));
    SGX_HIP(hipHostMalloc(&c->h_look, SGX_LOOK_BYTES, hipHostMallocCoherent | hipHostMallocMapped));
    memset(c->h_look, 0, SGX_LOOK_BYTES);
    SGX_HIP(hipHostGetDevicePointer(&c->d_look, c->h_look, 0));
    return SGX_OK;
}

extern "C" int sgx_ctx_create_prio(const sgx_settings* s, int device, int priority, sgx_ctx** out) {
    SGX_CHECK_ARG(s && out && priority >= -1 && priority <= 1);
    SGX_CHECK_ARG(s->codeLength == 1023 && s->samplingFreq > 0 && s->codeFreqBasis > 0);
    SGX_HIP(hipSetDevice(device));
    sgx_ctx* c = new sgx_ctx();
    c->s = *s;
    c->device = device;
    c->n_code = sgx_host_samples_per_code(s);
    memset(&c->timing, 0, sizeof(c->timing));
    c->priority = priority;
    const int rc = ctx_build(c, priority);
    if (rc != SGX_OK) {
        // a partially built context: release what exists (the message of the failing call is kept)
        if (c->stream) hipStreamDestroy(c->stream);
        for (int i = 0; i < 6; ++i)
            if (c->ev[i]) hipEventDestroy(c->ev[i]);
        if (c->d_codes) hipFree(c->d_codes);
        if (c->d_small) hipFree(c->d_small);
        if (c->h_small) hipHostFree(c->h_small);
        if (c->h_look) hipHostFree(c->h_look);
        delete c;
        return rc;
    }
    *out = c;
    return SGX_OK;
}

extern "C" int sgx_ctx_destroy(sgx_ctx* c) {
    if (!c) return SGX_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    sgx_fft_plan_destroy(&c->plan_code);
    sgx_fft_plan_destroy(&c->plan_fine);
    sgx_fft_plan_destroy(&c->plan_probe);
    for (int i = 0; i < 2; ++i)
        if (c->stage[i]) hipHostFree(c->stage[i]);
    hipFree(c->d_codes);
    if (c->d_sig64) hipFree(c->d_sig64);
    hipFree(c->d_fwd);
    hipFree(c->d_codefd);
    hipFree(c->d_work[0]);
    hipFree(c->d_work[1]);
    hipFree(c->d_pow);
    hipFree(c->d_fine[0]);
    hipFree(c->d_fine[1]);
    hipFree(c->d_small);
    hipFree(c->d_trk_out);
    hipFree(c->d_trk_aux);
    if (c->spare_d) hipFree(c->spare_d);
    if (c->spare_mark) hipFree(c->spare_mark);
    if (c->spare_copy_stream) hipStreamDestroy(c->spare_copy_stream);
    if (c->acq_stream2) hipStreamDestroy(c->acq_stream2);
    for (int i = 0; i < 2; ++i)
        if (c->acq_ev2[i]) hipEventDestroy(c->acq_ev2[i]);
    if (c->h_small) hipHostFree(c->h_small);
    if (c->h_look) hipHostFree(c->h_look);
    for (int i = 0; i < 6; ++i)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return SGX_OK;
}

extern "C" int sgx_ctx_sync(sgx_ctx* c) {
    SGX_CHECK_ARG(c);
    SGX_HIP(hipSetDevice(c->device));
    SGX_HIP(hipStreamSynchronize(c->stream));
    return SGX_OK;
}

extern "C" int sgx_get_timing(sgx_ctx* c, sgx_timing* out) {
    SGX_CHECK_ARG(c && out);
    *out = c->timing;
    return SGX_OK;
}

// pinned host memory for result buffers (D2H at full PCIe rate); plain C pointers, caller frees
extern "C" int sgx_host_alloc(size_t bytes, void** out) {
    SGX_CHECK_ARG(out && bytes > 0);
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        sgx_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return SGX_E_NOMEM;
    }
    *out = p;
    return SGX_OK;
}

extern "C" int sgx_host_free(void* p) {
    if (p) hipHostFree(p);
    return SGX_OK;
}

// ---- IF records ------------------------------------------------------------------------------

static int if_alloc(sgx_ctx* c, size_t n, sgx_if** out) {
    sgx_if* r = new sgx_if();
    r->n = n;
    r->device = c->device;
    hipError_t e = hipSuccess;
    {
        // the allocation the last freed record left behind, when it is large enough (and not absurdly larger)
        std::lock_guard<std::mutex> g(c->spare_mu);
        if (c->spare_d && c->spare_cap >= n + SGX_IF_PAD && c->spare_cap <= 2 * (n + SGX_IF_PAD) + (1u << 20)) {
            r->d = c->spare_d;
            r->cap = c->spare_cap;
            c->spare_d = nullptr;
            c->spare_cap = 0;
        }
    }
    if (!r->d) {
        e = hipMalloc((void**)&r->d, n + SGX_IF_PAD);
        if (e != hipSuccess) {
            // the parked allocation of an earlier record may be what is in the way: give it back and try once more
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g(c->spare_mu);
            if (c->spare_d) {
                hipFree(c->spare_d);
                c->spare_d = nullptr;
                c->spare_cap = 0;
                e = hipMalloc((void**)&r->d, n + SGX_IF_PAD);
            }
        }
        r->cap = n + SGX_IF_PAD;
    }
    if (e != hipSuccess) {
        delete r;
        sgx_set_error("hipMalloc(%zu) for an IF record failed: %s", n + SGX_IF_PAD, hipGetErrorString(e));
        return SGX_E_NOMEM;
    }
    e = hipMemsetAsync(r->d + n, 0, SGX_IF_PAD, c->stream);
    if (e != hipSuccess) {
        hipFree(r->d);
        delete r;
        sgx_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    *out = r;
    return SGX_OK;
}

int sgx_if_alloc_internal(sgx_ctx* c, size_t n, sgx_if** out) { return if_alloc(c, n, out); }

extern "C" int sgx_if_upload(sgx_ctx* c, const int8_t* host, size_t n, sgx_if** out) {
    SGX_CHECK_ARG(c && out && (host || n == 0));
    SGX_HIP(hipSetDevice(c->device));
    sgx_if* r = nullptr;
    int rc = if_alloc(c, n, &r);
    if (rc != SGX_OK) return rc;
    if (n) {
        hipError_t e = hipMemcpyAsync(r->d, host, n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // caller may free `host` on return
        if (e != hipSuccess) {
            hipFree(r->d);
            delete r;
            sgx_set_error("H2D copy of the IF record failed: %s", hipGetErrorString(e));
            return SGX_E_HIP;
        }
    }
    *out = r;
    return SGX_OK;
}

// ---- file -> HBM pipeline ------------------------------------------------------
// np.fromfile copies the file through the page cache into a pageable array and hipMemcpy then stages that array once
// more.  Here READERS threads pread() alternate 16 MiB chunks straight into a ring of four pinned slots while the
// issuing thread queues the slots' H2D copies in file order on one stream; a slot is read into again once the copy that
// last used it has completed.  One pread() stream moves ~21 GB/s out of the page cache (it is a CPU memcpy), three keep
// ahead of the PCIe link.
__global__ void if_mark_kernel(unsigned long long* mark, unsigned long long value) {
    __hip_atomic_store(mark, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

#define SGX_STAGE_BYTES (32u << 20)   // a pinned staging buffer: two slots
#define SGX_SLOT_BYTES (16u << 20)    // a multiple of every cache-line size: a line is never fetched half written
#ifndef SGX_PIPE_SLOTS
#define SGX_PIPE_SLOTS 4
#endif
#ifndef SGX_PIPE_READERS
#define SGX_PIPE_READERS 3
#endif

struct FilePipe {
    int fd = -1;
    uint64_t file_offset = 0;
    size_t n = 0;                         // bytes to move
    int8_t* dst = nullptr;                // device
    int device = 0;
    hipStream_t stream = nullptr;
    char* slot[SGX_PIPE_SLOTS] = {};
    hipEvent_t ev[SGX_PIPE_SLOTS] = {};
    unsigned long long* d_mark = nullptr; // device watermark advanced in stream order after every chunk, or null
    std::atomic<size_t>* host_mark = nullptr;   // bytes whose copy is known to have completed, or null
    std::vector<std::atomic<int>> read_ok;      // per chunk: 1 read, -1 read error
    std::atomic<long> issued{0};          // chunks whose copy and event have been queued
    std::atomic<bool> stop{false};
    std::atomic<int> err_no{0};
    std::atomic<size_t> err_off{0};
    explicit FilePipe(size_t chunks) : read_ok(chunks) {
        for (auto& f : read_ok) f.store(0);
    }
};

static const char* pipe_io_text(int err_no) {
    return err_no ? strerror(err_no) : "the file ends there (truncated while it was read?)";
}

static void pipe_reader(FilePipe* P, int t) {
    (void)hipSetDevice(P->device);
    const long chunks = (long)P->read_ok.size();
    for (long i = t; i < chunks && !P->stop.load(); i += SGX_PIPE_READERS) {
        const int sl = (int)(i % SGX_PIPE_SLOTS);
        if (i >= SGX_PIPE_SLOTS) {
            // the slot's previous chunk: its copy must have been queued, then completed
            while (P->issued.load() <= i - SGX_PIPE_SLOTS && !P->stop.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if (P->stop.load()) break;
            if (hipEventSynchronize(P->ev[sl]) != hipSuccess) {
                P->read_ok[(size_t)i].store(-1);
                break;
            }
            if (P->host_mark) {
                const size_t end = (size_t)(i - SGX_PIPE_SLOTS + 1) * SGX_SLOT_BYTES;
                size_t cur = P->host_mark->load();
                while (end > cur && !P->host_mark->compare_exchange_weak(cur, end)) {
                }
            }
        }
        const size_t off = (size_t)i * SGX_SLOT_BYTES;
        const size_t len = (P->n - off < SGX_SLOT_BYTES) ? (P->n - off) : SGX_SLOT_BYTES;
        size_t got = 0;
        bool bad = false;
        while (got < len) {
            const ssize_t m = pread(P->fd, P->slot[sl] + got, len - got, (off_t)(P->file_offset + off + got));
            if (m <= 0) {
                bad = true;
                P->err_no.store(m == 0 ? 0 : errno);   // 0: the file ended here (it was truncated while streaming)
                P->err_off.store(off + got);
                break;
            }
            got += (size_t)m;
        }
        P->read_ok[(size_t)i].store(bad ? -1 : 1);
        if (bad) break;
    }
}

// Runs the pipeline to completion on the calling thread (which issues the copies).  Returns hipSuccess and *io_fail.
static hipError_t pipe_run(FilePipe* P, bool* io_fail) {
    *io_fail = false;
    hipError_t e = hipSuccess;
    for (int i = 0; i < SGX_PIPE_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&P->ev[i], hipEventDisableTiming);
    std::vector<std::thread> readers;
    const long chunks = (long)P->read_ok.size();
    if (e == hipSuccess)
        for (int t = 0; t < SGX_PIPE_READERS && t < chunks; ++t) readers.emplace_back(pipe_reader, P, t);
    // host_mark follows the copies chunk by chunk (not only when a slot is reused, 4 chunks later): the prefix an
    // acquisition waits for is released as soon as its copy has completed.  Chunks up to issued - SLOTS are complete
    // (their slot has been refilled, which waits for their event); the events of the later ones are still their own.
    long completed = 0;
    auto advance = [&](long issued) {
        if (!P->host_mark) return;
        if (completed < issued - SGX_PIPE_SLOTS) completed = issued - SGX_PIPE_SLOTS;
        while (completed < issued && hipEventQuery(P->ev[completed % SGX_PIPE_SLOTS]) == hipSuccess) ++completed;
        size_t end = (size_t)completed * SGX_SLOT_BYTES;
        if (end > P->n) end = P->n;
        size_t cur = P->host_mark->load();
        while (end > cur && !P->host_mark->compare_exchange_weak(cur, end)) {
        }
    };
    for (long i = 0; i < chunks && e == hipSuccess; ++i) {
        int st;
        while ((st = P->read_ok[(size_t)i].load()) == 0) {
            advance(i);
            std::this_thread::sleep_for(std::chrono::microseconds(10));
        }
        if (st < 0) {
            *io_fail = true;
            break;
        }
        const int sl = (int)(i % SGX_PIPE_SLOTS);
        const size_t off = (size_t)i * SGX_SLOT_BYTES;
        const size_t len = (P->n - off < SGX_SLOT_BYTES) ? (P->n - off) : SGX_SLOT_BYTES;
        e = hipMemcpyAsync(P->dst + off, P->slot[sl], len, hipMemcpyHostToDevice, P->stream);
        if (e == hipSuccess && P->d_mark) if_mark_kernel<<<1, 1, 0, P->stream>>>(P->d_mark, (unsigned long long)(off + len));
        if (e == hipSuccess) e = hipEventRecord(P->ev[sl], P->stream);
        if (e != hipSuccess) P->err_off.store(off);   // (the chunk whose copy could not be queued)
        P->issued.store(i + 1);
    }
    if (e != hipSuccess || *io_fail) P->stop.store(true);
    for (auto& t : readers) t.join();
    // the tail: chunk by chunk as well (a record of a few chunks is all tail)
    while (P->host_mark && e == hipSuccess && !*io_fail && completed < chunks) {
        const long before = completed;
        advance(chunks);
        if (completed == before && hipEventSynchronize(P->ev[completed % SGX_PIPE_SLOTS]) != hipSuccess) break;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(P->stream);
    for (int i = 0; i < SGX_PIPE_SLOTS; ++i)
        if (P->ev[i]) hipEventDestroy(P->ev[i]);
    return e;
}

// the context's two pinned staging buffers (kept between calls: pinning 64 MiB costs ~15 ms), reserved for one user
static bool stage_acquire(sgx_ctx* c) {
    bool expected = false;
    if (!c->stage_busy.compare_exchange_strong(expected, true)) return false;
    for (int i = 0; i < 2; ++i)
        if (!c->stage[i] && hipHostMalloc(&c->stage[i], SGX_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) c->stage[i] = nullptr;
    if (c->stage[0] && c->stage[1]) return true;
    c->stage_busy.store(false);
    return false;
}

static bool pipe_slots(FilePipe* P, sgx_ctx* owner, void* own[2]) {
    own[0] = own[1] = nullptr;
    for (int i = 0; i < 2; ++i) {
        void* buf = owner ? owner->stage[i] : nullptr;
        if (!buf) {
            if (hipHostMalloc(&own[i], SGX_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) return false;
            buf = own[i];
        }
        P->slot[2 * i] = (char*)buf;
        P->slot[2 * i + 1] = (char*)buf + SGX_SLOT_BYTES;
    }
    return true;
}

extern "C" int sgx_if_upload_file(sgx_ctx* c, const char* path, uint64_t file_offset, size_t n, sgx_if** out) {
    SGX_CHECK_ARG(c && path && out);
    SGX_HIP(hipSetDevice(c->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        sgx_set_error("cannot open %s: %s", path, strerror(errno));
        return SGX_E_ARG;
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        sgx_set_error("fstat(%s) failed: %s", path, strerror(errno));
        return SGX_E_ARG;
    }
    size_t avail = ((uint64_t)sb.st_size > file_offset) ? (size_t)((uint64_t)sb.st_size - file_offset) : 0;
    if (avail > n) avail = n;
    sgx_if* r = nullptr;
    int rc = if_alloc(c, avail, &r);
    if (rc != SGX_OK) {
        close(fd);
        return rc;
    }
    FilePipe P((avail + SGX_SLOT_BYTES - 1) / SGX_SLOT_BYTES);
    P.fd = fd;
    P.file_offset = file_offset;
    P.n = avail;
    P.dst = r->d;
    P.device = c->device;
    P.stream = c->stream;
    sgx_ctx* owner = stage_acquire(c) ? c : nullptr;
    void* own[2];
    bool io_fail = false;
    hipError_t e = pipe_slots(&P, owner, own) ? pipe_run(&P, &io_fail) : hipErrorOutOfMemory;
    for (int i = 0; i < 2; ++i)
        if (own[i]) hipHostFree(own[i]);
    if (owner) owner->stage_busy.store(false);
    close(fd);
    if (e != hipSuccess || io_fail) {
        sgx_if_free(c, r);
        if (io_fail)
            sgx_set_error("read error on %s at byte %llu: %s", path, (unsigned long long)(file_offset + P.err_off.load()),
                          pipe_io_text(P.err_no.load()));
        else
            sgx_set_error("streaming upload of %s failed: %s", path, hipGetErrorString(e));
        return io_fail ? SGX_E_ARG : SGX_E_HIP;
    }
    *out = r;
    return SGX_OK;
}

// ---- background streaming: the record fills in file order while acquisition and tracking already run ----------
static void if_loader_main(sgx_if* r, int fd, uint64_t file_offset, std::string path, sgx_ctx* owner) {
    hipError_t e = hipSetDevice(r->device);
    FilePipe P((r->n + SGX_SLOT_BYTES - 1) / SGX_SLOT_BYTES);
    P.fd = fd;
    P.file_offset = file_offset;
    P.n = r->n;
    P.dst = r->d;
    P.device = r->device;
    P.stream = r->copy_stream;
    P.d_mark = r->d_mark;
    P.host_mark = &r->host_mark;
    void* own[2] = {nullptr, nullptr};
    bool io_fail = false;
    if (e == hipSuccess) e = pipe_slots(&P, owner, own) ? pipe_run(&P, &io_fail) : hipErrorOutOfMemory;
    if (e != hipSuccess || io_fail) {
        snprintf(r->load_err, sizeof(r->load_err), io_fail ? "read error on %s at byte %llu: %s" : "streaming %s failed at byte %llu: %s",
                 path.c_str(), (unsigned long long)(file_offset + P.err_off.load()),
                 io_fail ? pipe_io_text(P.err_no.load()) : hipGetErrorString(e));
        r->load_rc.store(io_fail ? SGX_E_ARG : SGX_E_HIP);
    } else {
        r->host_mark.store(r->n);
    }
    // whatever happened, nobody may wait for the watermark any longer
    if_mark_kernel<<<1, 1, 0, r->copy_stream>>>(r->d_mark, 0x7FFFFFFFFFFFFFFFull);
    hipStreamSynchronize(r->copy_stream);
    for (int i = 0; i < 2; ++i)
        if (own[i]) hipHostFree(own[i]);
    if (owner) owner->stage_busy.store(false);
    close(fd);
    r->load_done.store(true);
}

int sgx_if_require(const sgx_if* r, size_t end) {
    if (!r->loader) return SGX_OK;
    if (end > r->n) end = r->n;
    while (!r->load_done.load() && r->host_mark.load() < end) std::this_thread::sleep_for(std::chrono::microseconds(50));
    const int rc = r->load_rc.load();
    if (rc != SGX_OK) sgx_set_error("%s", r->load_err);
    return rc;
}

extern "C" int sgx_if_open_file(sgx_ctx* c, const char* path, uint64_t file_offset, size_t n, sgx_if** out) {
    SGX_CHECK_ARG(c && path && out);
    SGX_HIP(hipSetDevice(c->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        sgx_set_error("cannot open %s: %s", path, strerror(errno));
        return SGX_E_ARG;
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        sgx_set_error("fstat(%s) failed: %s", path, strerror(errno));
        return SGX_E_ARG;
    }
    size_t avail = ((uint64_t)sb.st_size > file_offset) ? (size_t)((uint64_t)sb.st_size - file_offset) : 0;
    if (avail > n) avail = n;
    sgx_if* r = nullptr;
    int rc = if_alloc(c, avail, &r);
    if (rc != SGX_OK) {
        close(fd);
        return rc;
    }
    hipError_t e = hipStreamSynchronize(c->stream);   // the zero pad is in place
    if (e == hipSuccess) {
        // A stream of the highest priority: HIP keeps separate hardware queues per priority, so the copies and the
        // watermark updates never queue up behind the (normal-priority) stream that runs the tracking kernel.
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* pe = getenv("SGX_STREAM_PRIO");   // test hook: "0" = a normal-priority copy stream
        if (pe && pe[0] == '0') hi = 0;
        if (c->priority < 0) hi = 0;   // the context itself runs at the highest priority: copies go one level below
        {
            std::lock_guard<std::mutex> g(c->spare_mu);
            if (c->spare_copy_stream && !(pe && pe[0] == '0')) {
                r->copy_stream = c->spare_copy_stream;
                c->spare_copy_stream = nullptr;
            }
            if (c->spare_mark) {
                r->d_mark = c->spare_mark;
                c->spare_mark = nullptr;
            }
        }
        if (e == hipSuccess && !r->copy_stream) e = hipStreamCreateWithPriority(&r->copy_stream, hipStreamNonBlocking, hi);
        if (e != hipSuccess) {   // no stream priorities here: an ordinary stream (the kernel's bounded wait covers it)
            (void)hipGetLastError();
            e = hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking);
        }
    }
    if (e == hipSuccess && !r->d_mark) e = hipMalloc((void**)&r->d_mark, 256);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_mark, 0, 256, r->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->copy_stream);
    if (e != hipSuccess) {
        close(fd);
        sgx_if_free(c, r);
        sgx_set_error("cannot set up the streaming record: %s", hipGetErrorString(e));
        return SGX_E_HIP;
    }
    // reserve the context's pinned staging buffers for this loader if nobody else is streaming
    sgx_ctx* owner = stage_acquire(c) ? c : nullptr;
    r->loader = new std::thread(if_loader_main, r, fd, file_offset, std::string(path), owner);
    *out = r;
    return SGX_OK;
}

extern "C" int sgx_if_wait(sgx_ctx* c, sgx_if* r, size_t n) {
    SGX_CHECK_ARG(c && r);
    return sgx_if_require(r, n == 0 ? r->n : n);
}

extern "C" int sgx_if_download(sgx_ctx* c, const sgx_if* r, size_t offset, size_t n, int8_t* host) {
    SGX_CHECK_ARG(c && r && host);
    SGX_CHECK_ARG(offset <= r->n && n <= r->n - offset);
    {
        const int rq = sgx_if_require(r, offset + n);
        if (rq != SGX_OK) return rq;
    }
    SGX_HIP(hipSetDevice(c->device));
    SGX_HIP(hipMemcpyAsync(host, r->d + offset, n, hipMemcpyDeviceToHost, c->stream));
    SGX_HIP(hipStreamSynchronize(c->stream));
    return SGX_OK;
}

extern "C" int sgx_trk_math_eval(int32_t fn, double a, double b, double* out) {
    SGX_CHECK_ARG(out && fn >= 0 && fn <= 10);
    switch (fn) {
        case 6: out[0] = sgx_div1(a, b); break;
        case 7: out[0] = sgx_sqrt1(a); break;
        case 8: out[0] = sgx_atan_ratio_k(a, b, sgx_atan_coef()); break;
        case 9: sgx_rot_small(a, sgx_rot_coef(), out[0], out[1]); break;
        case 10: {   // a = 1023 - rem, b = codeFreq, at fs = 38.192 MHz: block length, and step_a in out[1]
            double inv_step;
            out[0] = (double)sgx_block_length(a, b, 38192000.0, 1.0 / 38192000.0, out[1], inv_step);
            break;
        }
        case 0: out[0] = sgx_fast_rcp(a); break;
        case 1: out[0] = sgx_fast_div(a, b); break;
        case 2: out[0] = sgx_fast_sqrt(a); break;
        case 3: out[0] = sgx_atan_ratio(a, b); break;
        case 4: sgx_sincos_turns_short(a, out[0], out[1]); break;
        default: out[0] = (double)sgx_ceil_div(a, b); break;
    }
    return SGX_OK;
}

// ---- co-residency budget of cooperative tracking launches ---------------------------------------
static std::atomic<int> g_cu_used[64];

int sgx_cu_reserve(int device, int cus_total, int want) {
    if (device < 0 || device >= 64 || want <= 0) return 0;
    int cur = g_cu_used[device].load();
    for (;;) {
        if (cur + want > cus_total) return 0;
        if (g_cu_used[device].compare_exchange_weak(cur, cur + want)) return want;
    }
}

void sgx_cu_release(int device, int n) {
    if (device >= 0 && device < 64 && n > 0) g_cu_used[device].fetch_sub(n);
}

extern "C" int sgx_if_length(const sgx_if* r, size_t* n) {
    SGX_CHECK_ARG(r && n);
    *n = r->n;
    return SGX_OK;
}

extern "C" int sgx_if_free(sgx_ctx* c, sgx_if* r) {
    if (!r) return SGX_OK;
    if (r->loader) {
        r->loader->join();
        delete r->loader;
        r->loader = nullptr;
    }
    if (c) {
        hipSetDevice(c->device);
        hipStreamSynchronize(c->stream);
    }
    if (r->copy_stream) hipStreamSynchronize(r->copy_stream);
    if (c) {
        // keep ONE allocation (the larger), watermark and copy stream for the next record of this context
        std::lock_guard<std::mutex> g(c->spare_mu);
        const char* sp = getenv("SGX_IF_SPARE");   // '0': nothing is parked, a freed record's memory goes back at once
        if (r->d && r->cap > c->spare_cap && !(sp && sp[0] == '0')) {
            if (c->spare_d) hipFree(c->spare_d);
            c->spare_d = r->d;
            c->spare_cap = r->cap;
            r->d = nullptr;
        }
        if (r->d_mark && !c->spare_mark) {
            c->spare_mark = r->d_mark;
            r->d_mark = nullptr;
        }
        if (r->copy_stream && !c->spare_copy_stream && !getenv("SGX_STREAM_PRIO")) {
            c->spare_copy_stream = r->copy_stream;
            r->copy_stream = nullptr;
        }
    }
    if (r->copy_stream) hipStreamDestroy(r->copy_stream);
    if (r->d_mark) hipFree(r->d_mark);
    if (r->d) hipFree(r->d);
    delete r;
    return SGX_OK;
}

// ---- RCCL peak gather -------------------------------------------------------------------------
// librccl is opened lazily so that the library loads (and the host helpers work) on machines
// without a GPU.

struct RcclUid {
    char internal[128];
};
typedef int (*fn_get_uid)(RcclUid*);
typedef int (*fn_init_rank)(void**, int, RcclUid, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

static struct {
    void* h;
    fn_get_uid get_uid;
    fn_init_rank init_rank;
    fn_allgather allgather;
    fn_destroy destroy;
    fn_errstr errstr;
} g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

static int rccl_load() {
    if (g_rccl.h) return SGX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        sgx_set_error("cannot dlopen librccl: %s", dlerror());
        return SGX_E_RCCL;
    }
    g_rccl.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.allgather || !g_rccl.destroy) {
        sgx_set_error("librccl lacks an expected symbol");
        dlclose(h);
        return SGX_E_RCCL;
    }
    g_rccl.h = h;
    return SGX_OK;
}

// (struct sgx_comm: sgx_internal.h)

static int rccl_fail(const char* what, int code) {
    sgx_set_error("%s failed: %s", what, g_rccl.errstr ? g_rccl.errstr(code) : "rccl error");
    return SGX_E_RCCL;
}

extern "C" int sgx_comm_unique_id(uint8_t id[128]) {
    SGX_CHECK_ARG(id);
    int rc = rccl_load();
    if (rc != SGX_OK) return rc;
    RcclUid u;
    int e = g_rccl.get_uid(&u);
    if (e != 0) return rccl_fail("ncclGetUniqueId", e);
    memcpy(id, u.internal, 128);
    return SGX_OK;
}

extern "C" int sgx_comm_create(sgx_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t id[128],
                               sgx_comm** out) {
    SGX_CHECK_ARG(c && id && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks);
    int rc = rccl_load();
    if (rc != SGX_OK) return rc;
    SGX_HIP(hipSetDevice(c->device));
    RcclUid u;
    memcpy(u.internal, id, 128);
    void* comm = nullptr;
    int e = g_rccl.init_rank(&comm, n_ranks, u, rank);
    if (e != 0) return rccl_fail("ncclCommInitRank", e);
    sgx_comm* m = new sgx_comm();
    m->ctx = c;
    m->comm = comm;
    m->n_ranks = n_ranks;
    m->rank = rank;
    m->cap = 1 << 16;
    SGX_HIP(hipMalloc(&m->d_send, m->cap));
    SGX_HIP(hipMalloc(&m->d_recv, m->cap * (size_t)n_ranks));
    *out = m;
    return SGX_OK;
}

extern "C" int sgx_comm_allgather(sgx_comm* m, const void* send, void* recv, size_t bytes) {
    SGX_CHECK_ARG(m && send && recv && bytes > 0 && bytes <= m->cap);
    sgx_ctx* c = m->ctx;
    SGX_HIP(hipSetDevice(c->device));
    SGX_HIP(hipMemcpyAsync(m->d_send, send, bytes, hipMemcpyHostToDevice, c->stream));
    int e = g_rccl.allgather(m->d_send, m->d_recv, bytes, /*ncclInt8*/ 0, m->comm, c->stream);
    if (e != 0) return rccl_fail("ncclAllGather", e);
    SGX_HIP(hipMemcpyAsync(recv, m->d_recv, bytes * (size_t)m->n_ranks, hipMemcpyDeviceToHost, c->stream));
    SGX_HIP(hipStreamSynchronize(c->stream));
    return SGX_OK;
}

// ncclAllGather of `bytes` per rank from m->d_send into m->d_recv on the context's stream; nothing is copied or waited for
// (sgx_acquire_sharded packs and unpacks on the device)
int sgx_comm_allgather_device(sgx_comm* m, size_t bytes) {
    if (!m || bytes == 0 || bytes > m->cap) {
        sgx_set_error("sgx_comm_allgather_device: %zu bytes per rank, room for %zu", bytes, m ? m->cap : (size_t)0);
        return SGX_E_ARG;
    }
    int e = g_rccl.allgather(m->d_send, m->d_recv, bytes, /*ncclInt8*/ 0, m->comm, m->ctx->stream);
    if (e != 0) return rccl_fail("ncclAllGather", e);
    return SGX_OK;
}

extern "C" int sgx_comm_destroy(sgx_comm* m) {
    if (!m) return SGX_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    if (m->comm && g_rccl.destroy) g_rccl.destroy(m->comm);
    hipFree(m->d_send);
    hipFree(m->d_recv);
    delete m;
    return SGX_OK;
}
