#!/bin/bash
# (round 5) acquisition after a change: every acquisition test of the GPU suite, the two-rank test, the 4-PRN shard's
# timeline and the config-4 figures of the bench line.  GPU box: bash tools/r5_acq_check.sh
cd "$(dirname "$0")/.."
python -m pytest tests -q -m gpu -x -k "acqui or acquire or front_end or two_ranks or config4 or smoke" 2>&1 | tail -4
bash tools/acq_shard_trace.sh 0 2>&1 | tail -24
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --many-channels 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('acquire_ms', d['acquire_ms'], 'x_realtime', d['x_realtime'])
print({k:d['acq_config4'][k] for k in ('ms_n1','device_ms_n1','emulated_8rank_ms','emulated_speedup','sharded_result_equals_single_gpu')})"
