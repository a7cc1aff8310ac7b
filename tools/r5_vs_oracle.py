"""(round 5 diagnosis) the tracking kernel against the numpy oracle on the first `ms` blocks of the default scene: blocks whose recorded
code rate differs, the code phase difference they imply, envelope differences.  GPU box: python tools/r5_vs_oracle.py [ms]"""
import importlib, os, sys
import numpy as np
from concurrent.futures import ProcessPoolExecutor
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle_helpers import oracle_channel
m = importlib.import_module("softgnss-python_amd")
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
s = m.Settings(); s.msToProcess = float(ms); s.numberOfChannels = 8
ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
ser, dn = ctx.track(rec, chans, ms)
t = np.array(ser)
host = rec.download()
with ProcessPoolExecutor(max_workers=8) as ex:
    r = np.stack(list(ex.map(oracle_channel, [(host, p, f, c, ms) for p, f, c in chans])))
for ch in range(r.shape[0]):
    bad = np.nonzero(r[ch, 0] != t[ch, 0])[0]
    k = bad[0] if bad.size else ms
    dcf = (t[ch, 1] - r[ch, 1])[:k]
    blk = np.diff(np.concatenate([[r[ch, 0, 0] - 38192], r[ch, 0]]))[:k]
    drem = np.cumsum(blk[1:] * dcf[:-1] / 38.192e6) if k > 1 else np.zeros(1)
    env = lambda x, i, q: np.sqrt(x[ch, i, 200:k] ** 2 + x[ch, q, 200:k] ** 2)
    dE = env(t, 4, 6) / env(r, 4, 6) - 1; dL = env(t, 5, 8) / env(r, 5, 8) - 1
    print("ch %d: first boundary difference %s; codeFreq differs at %d of %d blocks; d(rem) max %.2e rms %.2e chips; d(codeNco) rms %.2e; E rms %.2e L rms %.2e E-L mean %.2e" %
          (ch, k if bad.size else None, np.count_nonzero(dcf), k, np.abs(drem).max(), drem.std(), (t[ch, 10] - r[ch, 10])[:k].std(), dE.std(), dL.std(), (dE - dL).mean()))

def rem_series(x, ch, k_end):
    """remCodePhase at the START of every block up to k_end, recomputed from the recorded block boundaries and code rates with
    the reference's own arithmetic (tracking.py:148-190)."""
    fs = 38.192e6
    first = x[ch, 0, 0] - 38192.0           # (the first block of the default front end is 38192 samples)
    pos = np.concatenate([[first], x[ch, 0]])
    rem = 0.0
    out = np.zeros(k_end + 1)
    cf = 1.023e6
    for k in range(k_end + 1):
        out[k] = rem
        blk = int(pos[k + 1] - pos[k])
        step = cf / fs
        stop = blk * step + rem
        stp = (stop - rem) / blk
        t_last = (blk - 1) * stp + rem
        rem = t_last + step - 1023.0
        cf = x[ch, 1, k]
    return out
kk = min(ms - 1, 9951)
for ch in (2, 7):
    rt, rr = rem_series(t, ch, kk), rem_series(r, ch, kk)
    d = rt - rr
    print("ch %d: rem(kernel) - rem(oracle) at block %d: %.3e chips (rem %.12e); max |.| over the blocks before %.3e, rms %.3e" % (ch, kk - 1, d[kk - 1], rr[kk - 1], np.abs(d[:kk]).max(), d[:kk].std()))
print("rem(kernel) - rem(oracle) in 1e-12 chips at blocks 1000, 2000, ...:")
for ch in range(r.shape[0]):
    rt, rr = rem_series(t, ch, kk), rem_series(r, ch, kk)
    print("  ch %d:" % ch, " ".join("%6.1f" % ((rt[k] - rr[k]) * 1e12) for k in range(1000, kk, 1000)))
