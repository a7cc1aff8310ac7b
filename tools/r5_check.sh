#!/bin/bash
# (round 5) quick correctness + timing of the tracking kernel on the GPU box:  bash tools/r5_check.sh [pytest -k expression]
cd "$(dirname "$0")/.."
out=gpurun_out/r5_check
mkdir -p $out
K="${1:-track_golden or track_device_file or replicated or full_length or split_variants or full_config3 or random_scenes or withheld or beyond_the_units or half_chip or other_front_ends or uint8}"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -${R5_TAIL:-8} $out/pytest.log
SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "profile\] ch 0 member\|^step" | awk 'NR<=3 || /member (9|10|19) / || /^step/' | head -8
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --many-channels 0 --concurrent 0 --no-config4 --no-from-file 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('track_kernel_ms %.3f  us/period %.4f  acquire_ms %.3f  x_realtime %.1f' % (d['track_kernel_ms'], d['us_per_code_period'], d['acquire_ms'], d['x_realtime']))"
