#!/bin/bash
# (round 5) acquisition times (config 2 and config 4, one GPU) with environment settings: bash tools/r5_acq_ab.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
for e in "" "$@"; do
  env $e python3 - <<PY
import importlib, os, sys
sys.path.insert(0, ".")
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 21 * n)
sig2 = m.DeviceSignal(rec, 0, 11 * n); sig4 = m.DeviceSignal(rec, 0, 20 * n)
def run(f):
    t = []
    for _ in range(8):
        a = f(); t.append(ctx.timing()["acquire_ms"])
    return min(t[2:]), a
t2, a2 = run(lambda: (lambda a: (a.acquire(sig2), a)[1])(m.AcquisitionResult(s, device=0)))
t4, a4 = run(lambda: (lambda a: (a.acquire(sig4, n_blocks=10, noncoh=True), a)[1])(m.AcquisitionResult(s, device=0)))
print("%-40s config 2: %.4f ms (%d det)   config 4: %.4f ms (%d det)  metric sum %.9f" % ("$e" or "(default)", t2, int((a2.carrFreq > 0).sum()), t4, int((a4.carrFreq > 0).sum()), float(a2.peakMetric.sum() + a4.peakMetric.sum())))
PY
done
