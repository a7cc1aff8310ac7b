"""(diagnosis) Many channels of an int16 / uint8 / int8 record on one GPU: channel-seconds per second and which kernel ran.
GPU box:  python3 tools/many_typed_probe.py [channels=1024] [ms=300]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ms = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
sc = m.synth.Scene.default()
span = 8 * n                                  # channels start up to 8 code periods apart
rec8 = ctx.synth(sc, m.synth.record_length(n, ms) + span + n).download()
a = m.AcquisitionResult(s, device=0)
a.acquire(rec8[:11 * n])
det = [i for i in range(32) if a.carrFreq[i] > 0][:8]
base = [(i + 1, float(a.carrFreq[i]), float(a.codePhase[i])) for i in det]
for name, arr, code in (("int8", rec8, m._native.DT_INT8),
                        ("uint8", (rec8.astype(np.int16) + 128).astype(np.uint8), m._native.DT_UINT8),
                        ("int16", rec8.astype("<i2") * 129 - 5, m._native.DT_INT16)):
    isz = arr.dtype.itemsize
    rec = ctx.upload_bytes(np.ascontiguousarray(arr).view(np.int8))
    # channel j: replica j % 8, started (j // 8) % 8 whole code periods later (the code phase repeats every period)
    chans = [(base[j % len(base)][0], base[j % len(base)][1],
              (base[j % len(base)][2] - 1 + ((j // 8) % 8) * n) * isz) for j in range(nch)]
    best = None
    for rep in range(3):
        ser, done = ctx.track(rec, chans, ms, data_type=code)
        t = ctx.timing()
        best = t["track_ms"] if best is None else min(best, t["track_ms"])
    ok = bool(np.all(done == ms))
    print("%-6s %5d channels x %d ms: kernel %8.2f ms  -> %9.0f channel-s/s   (track_kernel %d, members %d, all done %s)"
          % (name, nch, ms, best, nch * ms / best, t["track_kernel"], t["track_members"], ok))
    rec.free()
