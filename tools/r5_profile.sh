#!/bin/bash
# Round-5 profile run (GPU box): the rocprofv3 passes of tools/profile_round.sh, then the tracking kernel's phase times over the
# full run, the waves' barrier arrivals, the poll statistics and the slow-path counts -> gpurun_out/r05_phase.txt
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
V=$PWD/softgnss-python_amd/lib/variants
{
  echo "== per-member phase times, 37 000 blocks (SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000)"
  SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000 2>&1 | grep "profile\] ch 0 member\|^step"
  echo "== the round-4 kernel's phase times in the same run are in profiles/r04_trk_phase_profile.txt"
  echo "== barrier arrivals of the waves of (channel 0, unit 10), -DTRK_WAVEPROF build, 4000 blocks"
  SGX_LIB=$V/libsgx_wp.so python3 tools/step_profile.py 4000 2>&1 | grep "waveprof" | sort -u
  echo "== the filter waves by block parity (even blocks: set 0, the map waves on the filter waves' SIMDs, runs the final pass), 4000 blocks;"
  echo "== figures are per TWO blocks: double them.  -DT3_PROF_PAR=p: the PLL wave's view (publish, sums found, barrier released);"
  echo "== -DT3_PROF_DLL -DT3_PROF_PAR=p: the DLL wave's (release -> poll entered -> sums found -> at the barrier)"
  for v in pp0 pp1 pd0 pd1; do SGX_LIB=$V/libsgx_$v.so SGX_TRK_PROFILE=1 python3 tools/step_profile.py 4000 2>&1 | grep "ch 0 member 10 " | head -1 | sed "s/^/$v: /"; done
  echo "== poll statistics, -DT3_POLLSTAT build, 4000 blocks"
  SGX_LIB=$V/libsgx_ps.so python3 tools/step_profile.py 4000 2>&1 | grep "pollstat" | sort -u
  echo "== blocks off the plain path per wave, -DT3_COUNT build, 37 000 blocks"
  SGX_LIB=$V/libsgx_cnt.so python3 tools/step_profile.py 37000 2>&1 | grep "t3 count" | sort -u | awk '/unit  0 wave|unit  9 wave 0|unit 18 wave/'
} > gpurun_out/r05_phase.txt 2>&1
cat gpurun_out/r05_phase.txt | tail -60
ls gpurun_out/prof_r05/summary 2>/dev/null
