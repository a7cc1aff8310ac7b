#!/bin/bash
# (round 3) config 3 at full size against the oracle on the default scene and three random ones, then a fuzz sweep of
# random front ends.  GPU box; results -> gpurun_out/r03_full_parity.jsonl, gpurun_out/r03_fuzz.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r03_full_parity.jsonl
timeout 1500 python tools/full_parity.py 37000 2>&1 | tail -1 >> gpurun_out/r03_full_parity.jsonl
for seed in 7 8 9; do timeout 1500 python tools/full_parity.py 37000 $seed 2>&1 | tail -1 >> gpurun_out/r03_full_parity.jsonl; done
timeout 1500 python tools/fuzz_parity.py 100 160 2>&1 | tail -3 > gpurun_out/r03_fuzz.log
cat gpurun_out/r03_full_parity.jsonl gpurun_out/r03_fuzz.log
