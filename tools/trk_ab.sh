#!/bin/bash
# (diagnosis) tracking kernel time of the shipped library against variants (tools/build_variant.sh), alternating. GPU box:
#   bash tools/trk_ab.sh <reps> default <variant> ...
cd "$(dirname "$0")/.."
reps=$1; shift
for i in $(seq $reps); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset SGX_LIB; else export SGX_LIB=$PWD/softgnss-python_amd/lib/variants/libsgx_$v.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --many-channels 0 --concurrent 0 --no-config4 --no-from-file 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s track_kernel_ms %.3f  us/period %.4f  x_realtime %.1f' % ('$v', d['track_kernel_ms'], d['us_per_code_period'], d['x_realtime']))"
  done
done
