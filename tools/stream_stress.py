"""Streaming-record stress: many live streams (so stream -> hardware-queue assignments collide), then repeated
open_file + track cycles; every result must equal the resident run.  GPU box, run under `timeout`."""
import functools, importlib, os, sys, tempfile, time
import numpy as np
print = functools.partial(print, flush=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
ms = 8000
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
ref, done = ctx.track(rec, chans, ms)
path = os.path.join(tempfile.gettempdir(), "sgx_stress.bin")
rec.download().tofile(path)
size = os.path.getsize(path)
extra = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 13):
    s2 = m.Settings(); s2.acqThreshold = 2.5 + 0.01 * (k + 1)          # another settings key -> another context/stream
    extra.append(m._native.Context(s2, 0))
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    t0 = time.perf_counter()
    r = ctx.open_file(path, 0, size)
    ser, dn = ctx.track(r, chans, ms)
    dt = time.perf_counter() - t0
    ok = np.array_equal(ser, ref) and np.all(dn == ms)
    bad += (not ok)
    r.free()
    print("cycle %2d: %.1f ms, identical %s" % (it, dt * 1e3, ok))
print("stress done, mismatches:", bad)
os.remove(path)
