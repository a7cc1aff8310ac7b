#!/bin/bash
# (diagnosis) durations of the correlation kernels of BASELINE configs[3] (10 ms non-coherent). GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_var4
rocprofv3 --kernel-trace -d gpurun_out/prof_var4 -- python3 tools/acq_once4.py 4 > gpurun_out/var4.log 2>&1
grep acquire_ms gpurun_out/var4.log
python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var4/*/*_results.db")[0]
c = sqlite3.connect(db)
for pat in ("%fft4_cols%", "%fft4_rows%"):
    r = list(c.execute("select duration from kernels where name like ? order by start", (pat,)))
    big = sorted(x[0] for x in r if x[0] > 100000)
    print("%-14s n=%d min %7.1f med %7.1f us  (sum of the last call's %d: %.1f us)" % (pat, len(big), big[0] / 1e3, big[len(big) // 2] / 1e3, len(big) // 4, sum(big) / 4e3))
PY
python3 tools/acq_once4.py 4 2>&1 | tail -1
