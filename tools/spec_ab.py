"""(round 4 diagnosis) the speculative kernel (sgx_trk3.hip, SGX_TRK_V3=1) against the round-3 kernel (SGX_TRK_V3=0) on the same record:
first block whose boundary differs, largest difference of the sums before it.  GPU box: python tools/spec_ab.py [ms] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 37000
s = m.Settings(); s.msToProcess = float(ms); s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
out = {}
for spec in ("0", "1"):
    os.environ["SGX_TRK_V3"] = spec
    ser, dn = ctx.track(rec, chans, ms)
    out[spec] = np.array(ser)
    print("spec", spec, "kernel_ms %.3f" % ctx.timing()["track_ms"], "done", dn.tolist())
r, t = out["0"], out["1"]
scale = np.sqrt(np.mean(r[:, 3] ** 2 + r[:, 7] ** 2, axis=1))
for ch in range(r.shape[0]):
    bad = np.nonzero(r[ch, 0] != t[ch, 0])[0]
    k = bad[0] if bad.size else ms
    err = np.abs(t[ch, 3:9, :k] - r[ch, 3:9, :k]).max(axis=0) / scale[ch] if k else np.zeros(1)
    worst = int(np.argmax(err)) if k else -1
    print("ch %d PRN %2d: first boundary difference at block %s (of %d differing); sums before it: max rel %.3e at block %d, "
          "median %.3e; codeFreq %.3e Hz carrFreq %.3e Hz" % (ch, chans[ch][0], k if bad.size else None, bad.size, err.max(), worst,
          np.median(err), np.abs(t[ch, 1, :k] - r[ch, 1, :k]).max(), np.abs(t[ch, 2, :k] - r[ch, 2, :k]).max()))
    if bad.size:
        lo = max(0, k - 3)
        print("   blocks %d..%d  absoluteSample spec0 %s spec1 %s" % (lo, k + 1, r[ch, 0, lo:k + 2].tolist(), t[ch, 0, lo:k + 2].tolist()))
        print("   err around:", err[max(0, k - 6):k].tolist())
        big = np.nonzero(err > 1e-9)[0]
        print("   blocks with rel err > 1e-9 before it:", big[:20].tolist(), "count", big.size)
