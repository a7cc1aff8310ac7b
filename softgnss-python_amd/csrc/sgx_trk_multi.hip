// The cooperative tracking kernel for low sampling rates (fewer than ~15 samples per chip, e.g. 5.456 or 4.092
// Msps): a 16-sample group can then hold several chip switches of a code ramp, which the fast map of
// sgx_trk_kernel.inc excludes by construction, so this variant indexes the replicas sample by sample like the
// reference (tracking.py:166-188).  A translation unit of its own so that the default kernel's code generation
// is untouched (a run-time branch in the shared body cost the default kernel 11 %).
#include "sgx_trk_common.h"

#define TRK_MULTI 1
#define TRK_KERNEL_NAME trk_kernel_multi
#define TRK_MINW 1
#include "sgx_trk_kernel.inc"

void sgx_trk_multi_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                          double* out, int* ms_done, const TrkConst& K, long long* prof, unsigned long long* xch,
                          int* err) {
    trk_kernel_multi<<<n_blocks, TRK_THREADS, 0, st>>>(rec, codes, chans, out, ms_done, K, prof, xch, err);
}
