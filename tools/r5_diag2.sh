#!/bin/bash
# (round 5 diagnosis) barrier arrivals per wave by block parity; granule line layouts. GPU box.
cd "$(dirname "$0")/.."
out=gpurun_out/r5_diag2
mkdir -p $out
V=$PWD/softgnss-python_amd/lib/variants
for v in wp wp10; do
  echo "== $v"
  SGX_LIB=$V/libsgx_$v.so SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "^step\|waveprof" | sort
done > $out/wave.txt 2>&1
bash tools/trk_ab.sh 2 default xl20 xl24 > $out/ab.txt 2>&1
cat $out/wave.txt $out/ab.txt
