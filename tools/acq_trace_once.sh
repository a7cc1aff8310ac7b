#!/bin/bash
# Kernel-by-kernel timeline of ONE acquisition of BASELINE configs[1] (the last of N calls). GPU box: bash tools/acq_trace_once.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_acq1
rocprofv3 --kernel-trace -d gpurun_out/prof_acq1 -- python3 tools/acq_once.py 6 2>&1 | tail -2
python3 - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/prof_acq1/*/*_results.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
# the last call: everything after the last acq_mixphi_kernel... find the last 'acq_sum' start
idx = [i for i, r in enumerate(rows) if r[0].startswith("acq_sum")]
first = idx[-1]
t0 = rows[first][1]
prev_end = t0
for name, st, en, g, w in rows[first:]:
    print("%-58s start %8.1f us  dur %7.1f us  gap %6.1f  grid %d x %d" % (name[:58], (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, g // max(w, 1), w))
    prev_end = en
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
