#!/bin/bash
# (round 5 diagnosis) per-section cycles of the roles of trk3_kernel (member 0's... see T2STAMP), 4000 blocks, from a
# -DTRK_FINEPROF=1 variant: bash tools/build_variant.sh fp1 sgx_trk3.hip "-DTRK_FINEPROF=1"; (GPU box) bash tools/r5_fineprof.sh fp1
cd "$(dirname "$0")/.."
export SGX_LIB=$PWD/softgnss-python_amd/lib/variants/libsgx_${1:-fp1}.so
SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "fineprof\|^step" | sort | uniq -c | sort -k3,3 -k4,4n | head -60
