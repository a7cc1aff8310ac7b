"""(diagnosis) background file -> HBM streaming alone (open_file + wait) and beside the tracking kernel.  GPU box."""
import functools, importlib, os, sys, tempfile, time
print = functools.partial(print, flush=True)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
ms = 37000
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
path = os.path.join(tempfile.gettempdir(), "sgx_overlap.bin")
rec.download().tofile(path)
size = os.path.getsize(path)
for rep in range(3):
    t0 = time.perf_counter(); r = ctx.open_file(path, 0, size); r.wait(); t1 = time.perf_counter(); r.free()
    print("open_file + wait alone: %.1f ms" % ((t1 - t0) * 1e3))
for env in ({}, {"SGX_TRK_ARMS": "3"}, {"SGX_TRK_LDSPAD": "0"}):
    os.environ.update(env)
    for rep in range(2):
        t0 = time.perf_counter(); r = ctx.open_file(path, 0, size); ser, dn = ctx.track(r, chans, ms); t1 = time.perf_counter(); r.free()
        print("%s open_file + track: %.1f ms (kernel %.1f)" % (env, (t1 - t0) * 1e3, ctx.timing()["track_ms"]))
    for k in env: os.environ.pop(k)
os.remove(path)
