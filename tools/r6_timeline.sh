#!/bin/bash
# (round 6 diagnosis) builds the -DT3_TIMELINE variants of sgx_trk3.hip (one event of each role per build: a stamp costs
# ~60 cycles on the chain) - run HERE, before gpurun; then on the GPU box:  python3 tools/r6_timeline.py
cd "$(dirname "$0")/.."
for e in 1 2 3 4 5 6; do
  m=$((1 << e)); p=$m; d=$m
  if [ $e = 2 ]; then p=$((m | 64)); fi     # the PLL wave also reads the member's publish stamp where it finds the sums
  if [ $e -ge 5 ]; then p=0; d=0; fi
  bash tools/build_variant.sh tl$e sgx_trk3.hip "-DT3_TIMELINE -DT3_TL_MAP=$m -DT3_TL_PLL=$p -DT3_TL_DLL=$d" 2>&1 | grep -v warning | tail -2 &
done
wait
