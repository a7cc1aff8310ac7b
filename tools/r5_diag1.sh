#!/bin/bash
# (round 5 diagnosis) wave placement, phase times by block parity, fine probes, per-wave barrier arrivals of trk3_kernel. GPU box.
cd "$(dirname "$0")/.."
out=gpurun_out/r5_diag1
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 tools/ubench_place.hip -o /tmp/ubench_place && /tmp/ubench_place > $out/place.txt 2>&1
V=$PWD/softgnss-python_amd/lib/variants
for v in default par0 par1 fp1 wp; do
  echo "== $v"
  if [ $v = default ]; then unset SGX_LIB; else export SGX_LIB=$V/libsgx_$v.so; fi
  SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "profile\] ch 0 member\|^step\|fineprof\|waveprof" | awk '/member/ {n++; if (n<=3 || /member (9|10|19) /) print; next} {print}'
done > $out/variants.txt 2>&1
unset SGX_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "half_chip_spacing or track_golden or acquire_all_prns" > $out/pytest.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --many-channels 0 --concurrent 0 --no-config4 --no-from-file > $out/bench.json 2> $out/bench.err
cat $out/place.txt; cat $out/variants.txt; tail -3 $out/pytest.txt; python -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('x_realtime', d['x_realtime'], 'trk', d['track_kernel_ms'], 'acq', d['acquire_ms'])"
