#!/bin/bash
# (round 4 diagnosis) quick correctness + timing of the tracking kernel variants on the GPU box:
#   bash tools/r4_probe.sh [VAR=VALUE ...]    each argument is an environment setting of one timed variant
cd "$(dirname "$0")/.."
out=gpurun_out/r4_probe
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${R4_TESTS:-track_golden or track_device_file or replicated or full_length or split_variants or kernels_agree or full_config3 or random_scenes}" > $out/pytest1.log 2>&1
echo "pytest rc $?" >> $out/pytest1.log
tail -${R4_TAIL:-30} $out/pytest1.log
for v in "${@:-X=1}"; do
  echo "== variant [$v]"
  env $v SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "profile\] ch 0 member\|^step" | awk 'NR<=4 || /member  *(9|10|19|29) / || /^step/' | head -12
  env $v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --many-channels 0 --concurrent 0 --no-config4 --no-from-file 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('track_kernel_ms %.3f  us/period %.4f  acquire_ms %.3f  x_realtime %.1f' % (d['track_kernel_ms'], d['us_per_code_period'], d['acquire_ms'], d['x_realtime']))"
done 2>&1 | tee $out/variants.log
