"""(round 5 diagnosis) systematic differences between the speculative kernel and the round-3 kernel: mean and rms of the per-block
differences of the discriminators and of the three envelopes, per channel, over the blocks before the first boundary difference."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 9900
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
# series: 0 absoluteSample 1 codeFreq 2 carrFreq 3 I_P 4 I_E 5 I_L 6 Q_E 7 Q_P 8 Q_L 9 dllDiscr 10 dllDiscrFilt 11 pllDiscr 12 pllDiscrFilt
for ch in range(r.shape[0]):
    k0 = 200
    env = lambda x, i, q: np.sqrt(x[ch, i, k0:] ** 2 + x[ch, q, k0:] ** 2)
    dE = env(t, 4, 6) / env(r, 4, 6) - 1; dL = env(t, 5, 8) / env(r, 5, 8) - 1; dP = env(t, 3, 7) / env(r, 3, 7) - 1
    dd = t[ch, 9, k0:] - r[ch, 9, k0:]; dn_ = t[ch, 10, k0:] - r[ch, 10, k0:]
    dcar = np.diff(r[ch, 2])[k0 - 1:]
    eps2 = (2 * np.pi * dcar / 38.192e6) ** 2
    print("ch %d: d(dllDiscr) mean %.2e rms %.2e | d(codeNco) mean %.2e rms %.2e | E mean %.2e rms %.2e  L mean %.2e rms %.2e  P mean %.2e rms %.2e | E-L mean %.2e | corr(E-L, eps^2) %.3f" %
          (ch, dd.mean(), dd.std(), dn_.mean(), dn_.std(), dE.mean(), dE.std(), dL.mean(), dL.std(), dP.mean(), dP.std(), (dE - dL).mean(),
           np.corrcoef(dE - dL, eps2)[0, 1]))
print("code phase difference implied by the recorded code rates (chips): cumulative sum of blk * d(codeFreq) / fs")
for ch in range(r.shape[0]):
    blk = np.diff(np.concatenate([[r[ch, 0, 0] - 38192], r[ch, 0]]))
    dcf = t[ch, 1] - r[ch, 1]                    # codeFreq recorded at block k is the rate of block k + 1
    drem = np.cumsum(blk[1:] * dcf[:-1] / 38.192e6)
    nz = np.count_nonzero(dcf)
    print("ch %d: blocks with different codeFreq %d of %d; d(rem) max |.| %.2e chips, at the end %.2e, rms %.2e" % (ch, nz, dcf.size, np.abs(drem).max(), drem[-1], drem.std()))
