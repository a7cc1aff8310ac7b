// The cooperative tracking kernel for records of ANY sample type numpy reads (Settings.dataType; the reference's
// np.fromfile(fid, dataType, blksize), tracking.py:154): float32 / float64 records of arbitrary values, uint16 / int32 /
// ... records, and int16 / uint8 records at sampling rates the typed kernels exclude.  The per-sample body of
// sgx_trk_kernel.inc with every sample fetched where it lies, at any byte address (the reference seeks BYTES,
// tracking.py:107, so a channel may start inside a sample of the file - it does there, too), converted the way numpy's
// float64 arithmetic promotes it.  Compatibility before speed: ~5 us per code period for 8 channels (tools/any_type_probe.py).
#include "sgx_trk_common.h"

#define TRK_MULTI 1
#define TRK_ANY 1
#define TRK_KERNEL_NAME trk_kernel_any
#define TRK_MINW 1
#include "sgx_trk_kernel.inc"

void sgx_trk_any_launch(int n_blocks, hipStream_t st, const int8_t* rec, const int8_t* codes, const TrkChan* chans,
                        double* out, int* ms_done, const TrkConst& K, long long* prof, unsigned long long* xch,
                        int* err) {
    trk_kernel_any<<<n_blocks, TRK_THREADS, 0, st>>>(rec, codes, chans, out, ms_done, K, prof, xch, err);
}
