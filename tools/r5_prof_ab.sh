#!/bin/bash
# (round 5 diagnosis) phase profile (4000 blocks) of library variants: bash tools/r5_prof_ab.sh default NAME ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then unset SGX_LIB; else export SGX_LIB=$PWD/softgnss-python_amd/lib/variants/libsgx_$v.so; fi
  SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "ch 0 member 10 \|^step" | head -2 | sed "s/^/$v: /"
done
