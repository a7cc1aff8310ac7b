"""(round 6 diagnosis) the queued step against the eager one on the default scene: where the results differ, if they do."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s = m.Settings(); s.msToProcess = float(ms); s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
def step(deferred):
    a = m.AcquisitionResult(s, device=0, deferred=deferred); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
    t = m.TrackingResult(a, device=0); t.track(m.DeviceFile(rec)); return a, t
ae, te = step(False); ad, td = step(True); ad.results; ae2, te2 = step(False)
print("chained", td.chained, "eager twice equal", np.array_equal(te.series, te2.series))
print("channels eager", ae.channels.PRN, "deferred", ad.channels.PRN)
print("acqFreq equal", np.array_equal(ae.channels.acquiredFreq, ad.channels.acquiredFreq), "codePhase equal", np.array_equal(ae.channels.codePhase, ad.channels.codePhase))
d = np.abs(td.series - te.series)
print("nan eager", np.isnan(te.series).sum(), "nan deferred", np.isnan(td.series).sum(), "inf", np.isinf(te.series).sum(), np.isinf(td.series).sum())
for k, name in enumerate(m._native.SERIES):
    dk = d[:, k]
    bad = np.argwhere(dk > 0)
    print("%-16s max |diff| %.3e  first diff at %s" % (name, np.nanmax(dk), bad[0] if len(bad) else None))
