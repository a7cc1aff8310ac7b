#!/bin/bash
# (diagnosis) per-wave arrival times at the block barrier of the speculative tracking kernel (channel 0, member 0):
# a -DTRK_WAVEPROF build, one 4000-ms run.  GPU box: bash tools/waveprof.sh   -- restores the normal build afterwards
set -e
cd "$(dirname "$0")/.."
SGX_EXTRA_FLAGS="-DTRK_WAVEPROF ${WAVEPROF_EXTRA}" python softgnss-python_amd/build.py --force >/dev/null 2>&1
SGX_TRK_PROFILE=1 python tools/step_profile.py 4000 2>&1 | grep "waveprof\|member  [0129] \|member 1[08] \|^step"
python softgnss-python_amd/build.py --force >/dev/null 2>&1
