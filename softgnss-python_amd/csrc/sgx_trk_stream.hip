// The cooperative tracking kernel for a record that is still streaming into HBM (sgx_if_open_file): the body of
// sgx_trk_kernel.inc with the watermark checks compiled in.  A translation unit of its own, like the other
// variants, so that the default kernel's code generation is untouched.
#include "sgx_trk_common.h"

#define TRK_STREAM 1
#define TRK_KERNEL_NAME trk_kernel_stream
#define TRK_MINW 1
#include "sgx_trk_kernel.inc"

void sgx_trk_stream_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                           double* out, int* ms_done, const TrkConst& K, long long* prof, unsigned long long* xch,
                           int* err) {
    trk_kernel_stream<<<n_blocks, TRK_THREADS, 0, st>>>(rec, codes, chans, out, ms_done, K, prof, xch, err);
}
