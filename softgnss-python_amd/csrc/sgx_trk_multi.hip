// The cooperative tracking kernel for low sampling rates (fewer than ~15 samples per chip, e.g. 5.456 or 4.092
// Msps): a 16-sample group can then hold several chip switches of a code ramp, which the maps of sgx_trk2.hip
// exclude by construction, so this kernel indexes the replicas sample by sample like the reference
// (tracking.py:166-188).  int8 records; members own units c, c + split, ... (split <= 10).
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
