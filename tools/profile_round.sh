#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats of the bench command, FETCH_SIZE / WRITE_SIZE passes
# (separate, kernel trace only, as MI355X_MICROARCH.md prescribes) of the bench command and of three plain acquisitions,
# and the VALU passes of the many-channel tracking leg.  Usage (GPU box): bash tools/profile_round.sh r02
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
# the headline command alone: no concurrent-records or many-channel leg, so that trk2_kernel's average in the trace is the
# duration of the launches the bench line's roofline is computed from
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --many-channels 0 --no-from-file"
BENCH1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --concurrent 0 --no-config4 --no-from-file"
rocprofv3 --kernel-trace --stats -d $out/trace -- $BENCH > $out/trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -- $BENCH1 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -- $BENCH1 > /dev/null 2> $out/write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/acq_fetch -- python3 tools/acq_once.py 3 > /dev/null 2> $out/acq_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/acq_write -- python3 tools/acq_once.py 3 > /dev/null 2> $out/acq_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/valu -- $BENCH1 > /dev/null 2> $out/valu.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $out/grbm -- $BENCH1 > /dev/null 2> $out/grbm.err
python3 tools/profile_summary.py $tag $out
