"""Many-channel throughput mode: N channels (replicas of the 8 acquired ones) x ms on one GPU."""
import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
m = importlib.import_module('softgnss-python_amd')
s = m.Settings(); ctx = m.engine.get_context(s, 0)
sc = m.synth.Scene.default()
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rec = ctx.synth(sc, m.synth.record_length(s.samplesPerCode, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * s.samplesPerCode)); a.preRun()
ch8 = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
for n in (8, 256, 512, 1024, 2048, 4096):
    chans = [ch8[i % 8] for i in range(n)]
    ctx.track(rec, chans, 50)
    series, done = ctx.track(rec, chans, ms)
    t = ctx.timing()["track_ms"]
    b = n * ms * 38192.0 + n * ms * 13 * 8
    print("channels %4d  kernel %8.2f ms  %6.2f us/step  aggregate %7.1f GB/s (%.1f%% of 8 TB/s)  %.0f x real-time-channel" % (
        n, t, t * 1e3 / ms, b / t / 1e6, b / t / 1e6 / 80.0, n * ms / t))
