"""(round 6 diagnosis) the bench's step loop with a host-side split: where does the step's wall clock go beyond the two
device times?  python3 tools/step_wall3.py [nogc]"""
import gc, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("softgnss-python_amd")
shard = importlib.import_module("softgnss-python_amd.shard")
s = pkg.Settings(); s.msToProcess = 37000.0; s.numberOfChannels = 8
ctx = pkg.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(pkg.synth.Scene.default(), pkg.synth.record_length(n, 37000))
signal = pkg.DeviceSignal(rec, 0, 11 * n)
gather = shard.LocalGather()
last = {}
T = np.zeros(8)
def step():
    t0 = time.perf_counter()
    acq = pkg.AcquisitionResult(s, device=0, deferred=True)
    shard.acquire_sharded(acq, signal, 0, 1, gather)
    t1 = time.perf_counter()
    acq.preRun()
    trk = pkg.TrackingResult(acq, device=0)
    t2 = time.perf_counter()
    trk.track(pkg.DeviceFile(rec))
    t3 = time.perf_counter()
    acq.results
    t4 = time.perf_counter()
    last["acquire_ms"] = ctx.timing()["acquire_ms"]
    last["track_ms"] = trk.kernel_ms
    last["series"] = trk.series
    last["acq"] = acq
    t5 = time.perf_counter()
    T[:5] += (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)
    return trk
for _ in range(5): step()
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.collect(); gc.disable()
T[:] = 0
ctx.sync(); K = 20; dev = 0.0
t0 = time.perf_counter()
for _ in range(K):
    step(); dev += last["acquire_ms"] + last["track_ms"]
ctx.sync()
el = (time.perf_counter() - t0) / K * 1e3
print("step %.3f ms; device acquire + kernel %.3f ms; beyond: %.1f us | acquire() %.1f us, preRun + TrackingResult() %.1f, track() %.1f, acq.results %.1f, bookkeeping %.1f"
      % (el, dev / K, (el - dev / K) * 1e3, *(T[:5] / K * 1e6)))
