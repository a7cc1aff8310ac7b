"""Random front ends x sample types against the oracle: every seed draws a sampling rate, a scene, a Settings.dataType
(int8 ... float64), a scale / offset of the samples, the number of channels (2, or 160: throughput mode) and a start-byte
shift, tracks 30 ms and compares block boundaries (file bytes, exactly) and series (1e-9) with the oracle on the same
bytes.  Usage (GPU box): python tools/fuzz_types.py 0 60"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T

m = T.pkg()
orc = T.orc
lo, hi = int(sys.argv[1]), int(sys.argv[2])
FRONT = [(38192000.0, 9548000.0), (26000000.0, 6500000.0), (61380000.0, 15345000.0), (20460000.0, 5115000.0),
         (16368000.0, 4092000.0), (5456000.0, 1364000.0)]
TYPES = ["int8", "uint8", "int16", "uint16", "int32", "float32", "float64", "float16"]
bad = []
kernels = {}
for seed in range(lo, hi):
    rng = np.random.default_rng(0x7E5700 + seed)
    fs, IF = FRONT[int(rng.integers(len(FRONT)))]
    dtype = TYPES[int(rng.integers(len(TYPES)))]
    ms = 30
    s = m.Settings()
    os_ = orc.OracleSettings()
    nch = 2
    for o in (s, os_):
        o.samplingFreq, o.IF, o.numberOfChannels, o.msToProcess, o.dataType = fs, IF, nch, float(ms), dtype
    n = s.samplesPerCode
    prns = sorted(rng.choice(np.arange(1, 33), size=2, replace=False).tolist())
    dop = [float(rng.uniform(-5000, 5000)) for _ in prns]
    cph = [int(rng.integers(0, n)) for _ in prns]
    sc = m.synth.Scene.make(0x7E570000 + seed, fs, IF, prns, dop, cph, [int(rng.integers(6, 10)) for _ in prns])
    rec8 = m.synth.generate(sc, m.synth.record_length(n, ms) + n)
    x = rec8.astype(np.float64)
    if dtype == "int8":
        arr = rec8
    elif dtype == "uint8":
        arr = (rec8.astype(np.int16) + 128).astype(np.uint8)
    elif dtype == "int16":
        arr = (rec8.astype(np.int32) * int(rng.integers(1, 250)) + int(rng.integers(-50, 50))).astype("<i2")
    elif dtype == "uint16":
        arr = (rec8.astype(np.int32) * int(rng.integers(1, 200)) + 30000).astype("<u2")
    elif dtype == "int32":
        arr = (rec8.astype(np.int64) * int(rng.integers(1, 10 ** 6))).astype("<i4")
    elif dtype == "float32":
        arr = (x * float(rng.uniform(1e-4, 1e3)) + float(rng.uniform(-0.05, 0.05))).astype("<f4")
    elif dtype == "float64":
        arr = x * float(10.0 ** rng.uniform(-6, 6)) + float(rng.uniform(-1e-3, 1e-3))
    else:
        arr = (x * 0.125).astype("<f2")
    isz = arr.dtype.itemsize
    many = bool(rng.integers(0, 4) == 0)                 # one seed in four: 160 channels (throughput mode where it applies)
    shift = int(rng.integers(0, isz)) if (isz > 1 and dtype not in ("float32", "float64", "float16") and rng.integers(0, 3) == 0) else 0
    phase = np.array([c * isz + shift for c in cph], dtype=np.float64)
    freq = np.array([IF + d for d in dop])
    try:
        want = orc.stack_series(orc.track(os_, dict(PRN=np.array(prns), acquiredFreq=freq, codePhase=phase,
                                                    status=['T'] * 2), arr))
        ctx = m.engine.get_context(s, 0)
        chans = [(int(prns[i]), float(freq[i]), float(phase[i])) for i in range(2)]
        code = {"int8": 0, "int16": 1, "uint8": 2, "float32": 3, "float64": 4, "uint16": 5, "int32": 6, "float16": 10}[dtype]
        rec = ctx.upload_bytes(np.ascontiguousarray(arr).view(np.int8))
        got, done = ctx.track(rec, chans * (80 if many else 1), ms, data_type=code)
        rec.free()
        k = int(ctx.timing()["track_kernel"])
        kernels[(dtype, k)] = kernels.get((dtype, k), 0) + 1
        assert np.all(done == ms), "short"
        assert np.array_equal(got[:2, 0], want[:, 0]), "boundaries"
        e = T._trk_err(got[:2], want)
        assert e < T.TRK_TOL, "series %.3g" % e
        if many:
            assert all(np.array_equal(got[i], got[i % 2]) for i in range(2, 160)), "replicas differ"
    except Exception as ex:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, dtype, fs, "many" if many else "", "shift", shift, "FAILED:", repr(ex)[:200])
        traceback.print_exc(limit=1)
print("kernels run (dtype, track_kernel): %s" % sorted(kernels.items()))
print("seeds %d..%d: %d failures %s" % (lo, hi - 1, len(bad), bad))
