"""(round 6 diagnosis) wall-clock split of the queued step on the host: python3 tools/step_wall2.py [eager]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
eager = len(sys.argv) > 1 and sys.argv[1] == "eager"
s = m.Settings(); s.msToProcess = 37000.0; s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 37000))
sig = m.DeviceSignal(rec, 0, 11 * n)
def step(show):
    t = [time.perf_counter()]
    a = m.AcquisitionResult(s, device=0, deferred=not eager); a.acquire(sig); t.append(time.perf_counter())
    a.preRun(); t.append(time.perf_counter())
    tr = m.TrackingResult(a, device=0); t.append(time.perf_counter())
    if show: os.environ["SGX_STEP_TRACE"] = "1"
    tr.track(m.DeviceFile(rec)); t.append(time.perf_counter())
    os.environ.pop("SGX_STEP_TRACE", None)
    a.results; t.append(time.perf_counter())
    if show:
        tm = ctx.timing()
        print("acquire() %.1f us | preRun() %.1f | TrackingResult() %.1f | track() %.1f (kernel %.1f) | acq.results %.1f | step %.1f us; device: acquire %.1f + kernel %.1f = %.1f"
              % tuple([(t[i + 1] - t[i]) * 1e6 for i in range(3)] + [(t[4] - t[3]) * 1e6, tm["track_ms"] * 1e3, (t[5] - t[4]) * 1e6, (t[5] - t[0]) * 1e6,
                       tm["acquire_ms"] * 1e3, tm["track_ms"] * 1e3, (tm["acquire_ms"] + tm["track_ms"]) * 1e3]))
for i in range(6):
    step(i >= 4)
