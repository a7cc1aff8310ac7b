#!/bin/bash
# Round-6 profile run (GPU box, ~4 min): the rocprofv3 passes of tools/profile_round.sh (kernel trace + stats, FETCH_SIZE /
# WRITE_SIZE / VALU / GRBM passes), the tracking kernel's stamped phases over the full run, the default bench line.
# tools/r6_harvest.py then makes the tracked files under profiles/ (r06_*).
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
{
  echo "== per-member phase times, 37 000 blocks (SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000)"
  for i in 1 2 3; do SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000 2>&1 | grep "profile\] ch 0 member\|^step"; done
} > gpurun_out/r06_phase.txt 2>&1
python3 bench.py > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt
tail -1 gpurun_out/r06_bench_stdout.txt > gpurun_out/r06_bench_line.json
tail -3 gpurun_out/r06_phase.txt; ls gpurun_out/prof_r06/summary 2>/dev/null; cut -c1-600 gpurun_out/r06_bench_line.json
