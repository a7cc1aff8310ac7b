"""Wall-clock breakdown of one bench step (host side) vs HIP-event kernel times."""
import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
m = importlib.import_module('softgnss-python_amd')
s = m.Settings(); ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 37000))
sig = m.DeviceSignal(rec, 0, 11 * n)
for rep in range(3):
    t0 = time.perf_counter()
    a = m.AcquisitionResult(s, device=0); a.acquire(sig)
    t1 = time.perf_counter(); a.preRun()
    t2 = time.perf_counter(); t = m.TrackingResult(a, device=0)
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    t3 = time.perf_counter(); series, done = ctx.track(rec, chans, 37000)
    t4 = time.perf_counter(); t.track(m.DeviceFile(rec))
    t5 = time.perf_counter()
    tm = ctx.timing()
    print("acquire wall %.2f ms (device %.2f) | preRun %.2f | ctx.track wall %.2f ms (kernel %.2f) | TrackingResult.track wall %.2f ms" % (
        (t1 - t0) * 1e3, tm["acquire_ms"], (t2 - t1) * 1e3, (t4 - t3) * 1e3, tm["track_ms"], (t5 - t4) * 1e3))
