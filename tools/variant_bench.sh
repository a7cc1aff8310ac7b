#!/bin/bash
# Build the library with extra compile flags and print the tracking kernel time of a short bench run.
# Usage (GPU box): bash tools/variant_bench.sh "<flags A>" "<flags B>" ...   ("" = the normal build); restores the normal build
cd "$(dirname "$0")/.."
for flags in "$@"; do
    SGX_EXTRA_FLAGS="$flags" python softgnss-python_amd/build.py --force >/dev/null 2>&1
    for rep in 1 2; do
        python bench.py --steps 3 --warmup 1 --no-cpu-baseline --many-channels 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags [%s]: track_kernel_ms %.3f  us/period %.4f  acquire_ms %.3f  x_realtime %.1f' % ('$flags', d['track_kernel_ms'], d['us_per_code_period'], d['acquire_ms'], d['x_realtime']))"
    done
done
python softgnss-python_amd/build.py --force >/dev/null 2>&1
