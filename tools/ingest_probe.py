"""File -> HBM ingest rate: native streamer vs np.fromfile + upload, on a 1.4 GB record in /tmp."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, '.')
m = importlib.import_module('softgnss-python_amd')
s = m.Settings(); ctx = m.engine.get_context(s, 0)
n = 1413217385
path = '/tmp/sgx_ingest_probe.bin'
rec = ctx.synth(m.synth.Scene.default(), n)
rec.download().tofile(path); rec.free()
for name, fn in (("native stream", lambda: ctx.upload_file(path, 0, n)),
                 ("np.fromfile + upload", lambda: ctx.upload(np.fromfile(path, np.int8)))):
    for rep in range(2):
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        print("%-22s run %d: %.3f s  %.2f GB/s" % (name, rep, dt, n / dt / 1e9)); r.free()
os.remove(path)
