// Throughput-mode build of the cooperative tracking kernel: register allocation capped so that two
// 256-thread workgroups (two channels) share a CU.  Used when split == 1 and there are more channels than
// half the CUs.  Same source as sgx_trk.hip's trk_kernel (sgx_trk_kernel.inc).
#include "sgx_trk_common.h"

#define TRK_KERNEL_NAME trk_kernel_tp
#define TRK_MINW 2
#include "sgx_trk_kernel.inc"

void sgx_trk_tp_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const void* chans,
                       double* out, int* done, const TrkConst& K, long long* prof, unsigned long long* xch,
                       int* err) {
    trk_kernel_tp<<<n_blocks, TRK_THREADS, 0, st>>>(rec, codes, (const TrkChan*)chans, out, done, K, prof, xch, err);
}
