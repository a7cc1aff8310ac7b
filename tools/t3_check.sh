#!/bin/bash
# (diagnosis) -DT3_CHECK build of the speculative tracking kernel: every lane's six sums of the first blocks against the
# direct per-sample evaluation, mismatches printed.  GPU box: bash tools/t3_check.sh   -- restores the normal build afterwards
set -e
cd "$(dirname "$0")/.."
SGX_EXTRA_FLAGS="-DT3_CHECK" python softgnss-python_amd/build.py --force >/dev/null 2>&1
python tools/step_profile.py 40 2>&1 | grep "t3 check\|^step" | sort | head -${1:-40}
python softgnss-python_amd/build.py --force >/dev/null 2>&1
