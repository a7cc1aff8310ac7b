#!/bin/bash
# Diagnosis build of the tracking kernel with per-phase probes (see PROBE() in sgx_trk_common.h), one 4000-ms run.
# Usage (GPU box): bash tools/fineprof.sh     -- restores the normal build afterwards
set -e
cd "$(dirname "$0")/.."
SGX_EXTRA_FLAGS=-DTRK_FINEPROF=${FINEPROF_LEVEL:-1} python softgnss-python_amd/build.py --force >/dev/null 2>&1
SGX_TRK_PROFILE=1 python - <<'PY'
import importlib, sys
sys.path.insert(0, '.')
m = importlib.import_module('softgnss-python_amd')
s = m.Settings(); ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 4000))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
ctx.track(rec, chans, 4000)
series, done = ctx.track(rec, chans, 4000)
print("kernel ms", ctx.timing()["track_ms"], "us/block", ctx.timing()["track_ms"] * 1e3 / 4000)
PY
python softgnss-python_amd/build.py --force >/dev/null 2>&1
