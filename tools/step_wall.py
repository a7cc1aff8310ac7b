"""(diagnosis) wall time of the three calls of one headline step against their device times. GPU box."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
ms = 37000
s = m.Settings(); s.msToProcess = float(ms); s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
sig = m.DeviceSignal(rec, 0, 11 * n)
f = m.DeviceFile(rec)
acc = [0.0] * 6
N = 8
for k in range(N + 2):
    t0 = time.perf_counter(); a = m.AcquisitionResult(s, device=0); a.acquire(sig); t1 = time.perf_counter()
    dev_a = ctx.timing()["acquire_ms"]
    a.preRun(); t2 = time.perf_counter()
    t = m.TrackingResult(a, device=0); t.track(f); t3 = time.perf_counter()
    if k >= 2:
        for i, v in enumerate(((t1 - t0) * 1e3, dev_a, (t2 - t1) * 1e3, (t3 - t2) * 1e3, t.kernel_ms, (t3 - t0) * 1e3)): acc[i] += v / N
print("acquire wall %.3f ms (device %.3f)  preRun %.3f ms  track wall %.3f ms (kernel %.3f)  step %.3f ms" % tuple(acc))
