#!/bin/bash
# Kernel-by-kernel timeline of ONE acquisition of BASELINE configs[1] (the last of N calls). GPU box: bash tools/acq_trace_once.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_acq1
rocprofv3 --kernel-trace -d gpurun_out/prof_acq1 -- python3 tools/acq_once.py 6 2>&1 | tail -2
python3 - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/prof_acq1/*/*_results.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
first = [i for i, r in enumerate(rows) if r[0].startswith("acq_sum")][-1]
t0 = rows[first][1]; prev_end = t0
for name, st, en, g, w in rows[first:]:
    print("%-56s start %7.1f dur %6.1f gap %5.1f grid %5d x %3d" % (name[:56], (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, g // max(w, 1), w))
    prev_end = en
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
