#!/bin/bash
# rocprofv3 kernel-trace of a short bench run; prints the acquisition-side kernels. Usage (GPU box): bash tools/acq_profile.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_acq
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_acq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --many-channels 0 --concurrent 0 --ms 200 > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/prof_acq/*/*_results.db")[0]
c = sqlite3.connect(db)
for n, calls, tot, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if "trk_kernel" in n or "synth" in n or "stream_" in n:
        continue
    print("%-60s calls %3d  avg %8.1f us  per-step %8.1f us" % (n[:60], calls, avg, tot / 5))
PY
