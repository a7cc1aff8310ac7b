"""(round 5 diagnosis) new speculative kernel against the round-3 kernel: per-channel statistics of the carrier NCO steps and where the sums part."""
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
r, t = out["0"], out["1"]
scale = np.sqrt(np.mean(r[:, 3] ** 2 + r[:, 7] ** 2, axis=1))
for ch in range(r.shape[0]):
    dcar = np.diff(r[ch, 2]); dcode = np.diff(r[ch, 1])
    err = np.abs(t[ch, 3:9] - r[ch, 3:9]).max(axis=0) / scale[ch]
    big = np.nonzero(err > 1e-9)[0]
    print("ch %d PRN %2d scale %.0f: carr step rms %.2f Hz max %.1f Hz; code step rms %.3f Hz max %.2f; err>1e-9 at %d blocks, first %s" %
          (ch, chans[ch][0], scale[ch], dcar.std(), np.abs(dcar).max(), dcode.std(), np.abs(dcode).max(), big.size, big[:5].tolist()))
    if big.size:
        k = big[0]
        names = ["I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L"]
        print("   at block %d (new - old):" % k, ", ".join("%s %.4f (of %.1f)" % (names[i], t[ch, 3 + i, k] - r[ch, 3 + i, k], r[ch, 3 + i, k]) for i in range(6)),
              "| carrFreq %.3e codeFreq %.3e" % (t[ch, 2, k] - r[ch, 2, k], t[ch, 1, k] - r[ch, 1, k]))
        for j in range(max(0, k - 4), min(ms, k + 8)):
            print("   blk %d err %.2e  dcarr %.2f Hz dcode %.3f Hz  I_P %.1f / %.1f  absS %s" % (j, err[j], r[ch, 2, j] - r[ch, 2, j - 1], r[ch, 1, j] - r[ch, 1, j - 1], t[ch, 3, j], r[ch, 3, j], t[ch, 0, j] == r[ch, 0, j]))
