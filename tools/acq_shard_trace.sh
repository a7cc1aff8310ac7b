#!/bin/bash
# Kernel-by-kernel timeline of ONE 4-PRN shard of BASELINE configs[3] (10 ms non-coherent): what one rank of an 8-GPU
# run executes (the last of N calls).  GPU box: bash tools/acq_shard_trace.sh [first PRN index, default 0]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_shard
cat > /tmp/acq_shard_once.py <<PY
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 21 * n)
sig = m.DeviceSignal(rec, 0, 20 * n)
p0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(6):
    a = m.AcquisitionResult(s, device=0); a.acquire(sig, n_blocks=10, noncoh=True, prn_indices=list(range(p0, p0 + 4)))
print("acquire_ms", ctx.timing()["acquire_ms"], "detections", int((a.carrFreq > 0).sum()))
PY
rocprofv3 --kernel-trace -d gpurun_out/prof_shard -- python3 /tmp/acq_shard_once.py ${1:-0} 2>&1 | tail -2
python3 - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/prof_shard/*/*_results.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
first = [i for i, r in enumerate(rows) if r[0].startswith("acq_setup") or r[0].startswith("acq_front")][-1]
t0 = rows[first][1]; prev_end = t0
for name, st, en, g, w in rows[first:]:
    print("%-56s start %7.1f dur %6.1f gap %5.1f grid %5d x %3d" % (name[:56], (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, g // max(w, 1), w))
    prev_end = en
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
