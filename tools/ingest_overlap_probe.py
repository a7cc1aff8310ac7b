"""File-based run: record file -> HBM -> 8-channel x 37 s tracking, with the transfer (a) completed first
(sgx_if_upload_file) and (b) overlapped with the kernel (sgx_if_open_file + watermark).  GPU box."""
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
ref, done = ctx.track(rec, chans, ms)
print('reference run done')
path = os.path.join(tempfile.gettempdir(), "sgx_overlap.bin")
rec.download().tofile(path)
size = os.path.getsize(path)
rec.free()
for mode in ("upload_file", "open_file", "upload_file", "open_file"):
    t0 = time.perf_counter()
    r = getattr(ctx, mode)(path, 0, size)
    t1 = time.perf_counter()
    ser, dn = ctx.track(r, chans, ms)
    t2 = time.perf_counter()
    k_ms = ctx.timing()["track_ms"]
    ok = np.array_equal(ser, ref) and np.all(dn == ms)
    r.free()
    print("%-11s: ingest call %6.1f ms + track call %6.1f ms = %6.1f ms (kernel %.1f ms)  identical results: %s  -> %.0f Msamples/s from the file"
          % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, k_ms, ok, size / (t2 - t0) / 1e6))
os.remove(path)
