"""(diagnosis) float32 / float64 records of arbitrary values (trk2_kernel<4|8,3>; SGX_TRK_FLOAT_TYPED=0: trk_kernel_any): time per
code period for 8 channels (cooperating workgroups) and for 256 (one workgroup each).
GPU box:  python3 tools/any_type_probe.py [ms=2000]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")

ms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec8 = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms) + 9 * n).download()
a = m.AcquisitionResult(s, device=0)
a.acquire(rec8[:11 * n])
det = [i for i in range(32) if a.carrFreq[i] > 0][:8]
base = [(i + 1, float(a.carrFreq[i]), float(a.codePhase[i])) for i in det]
for name, arr, code in (("float32", (rec8.astype(np.float64) * 0.37 + 0.011).astype("<f4"), m._native.DT_FLOAT32),
                        ("float64", rec8.astype(np.float64) * 1.2345e-3, m._native.DT_FLOAT64)):
    isz = arr.dtype.itemsize
    rec = ctx.upload_bytes(np.ascontiguousarray(arr).view(np.int8))
    for nch in (8, 256):
        chans = [(base[j % len(base)][0], base[j % len(base)][1],
                  (base[j % len(base)][2] - 1 + ((j // 8) % 8) * n) * isz) for j in range(nch)]
        ser, done = ctx.track(rec, chans, ms, data_type=code)
        t = ctx.timing()
        print("%-8s %4d channels x %d ms: kernel %8.2f ms = %.2f us per code period, %.0f x real time per channel set, "
              "%.0f channel-s/s (track_kernel %d, members %d, done %s)"
              % (name, nch, ms, t["track_ms"], t["track_ms"] * 1e3 / ms, ms / t["track_ms"], nch * ms / t["track_ms"],
                 t["track_kernel"], t["track_members"], bool(np.all(done == ms))))
    rec.free()
