"""(diagnosis) host-side profile of one headline step: acquire -> preRun -> track of 8 channels x 37 000 ms. GPU box."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 37000
s = m.Settings(); s.msToProcess = float(ms); s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
sig = m.DeviceSignal(rec, 0, 11 * n)
def step():
    a = m.AcquisitionResult(s, device=0); a.acquire(sig); a.preRun()
    t = m.TrackingResult(a, device=0); t.track(m.DeviceFile(rec))
    return t
for _ in range(2): step()
t0 = time.perf_counter(); t = step(); dt = time.perf_counter() - t0
print("step %.3f ms, kernel %.3f ms, acquire %.3f ms" % (dt * 1e3, t.kernel_ms, ctx.timing()["acquire_ms"]))
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
