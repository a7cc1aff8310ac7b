import os, sys, traceback
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import softgnss_oracle as orc
seed = int(sys.argv[1])
m = T.pkg()
rng = np.random.default_rng(seed)
fs = float(rng.choice([38192000.0, 16367600.0, 26000000.0, 20460000.0, 12276000.0, 5456000.0]))
IF = float(rng.choice([0.25, 0.2, 0.31]) * fs)
s = m.Settings(); os_ = orc.OracleSettings()
kw = dict(samplingFreq=fs, IF=IF, acqSatelliteList=range(1, 6), numberOfChannels=2, msToProcess=30.0,
          dllCorrelatorSpacing=float(rng.choice([0.5, 0.5, 0.3, 0.7])), dllNoiseBandwidth=float(rng.choice([2.0, 1.0, 5.0])),
          pllNoiseBandwidth=float(rng.choice([25.0, 10.0, 50.0])), acqSearchBand=float(rng.choice([14.0, 14.0, 8.0])))
for k, v in kw.items():
    setattr(s, k, v); setattr(os_, k, v)
print(kw)
n = s.samplesPerCode
half = kw["acqSearchBand"] * 500.0
prns = sorted(rng.choice(np.arange(1, 6), size=2, replace=False).tolist())
dop = [float(rng.uniform(-half, half)) for _ in prns]; ph = [int(rng.integers(0, n)) for _ in prns]; amp = [int(rng.integers(4, 10)) for _ in prns]
print("n", n, "prns", prns, "doppler", dop, "phase", ph, "amp", amp)
sc = m.synth.Scene.make(0xAB000 + seed, fs, IF, prns, dop, ph, amp)
rec = m.synth.generate(sc, m.synth.record_length(n, 30))
a = m.AcquisitionResult(s, device=0); a.acquire(rec[:11 * n])
ref = orc.acquire(os_, rec[:11 * n])
for f in ("codePhase", "carrFreq", "peakMetric"):
    print(f, "gpu", a.results[f][:5], "oracle", ref[f][:5])
print("freqBin", a.internals["freqBin"][:5], "fineIdx", a.internals["fineIdx"][:5], "oracle", ref.get("freqBin", None) if isinstance(ref, dict) else None)
if not (np.array_equal(a.codePhase, ref["codePhase"]) and np.array_equal(a.carrFreq, ref["carrFreq"])):
    print("ACQUISITION DIFFERS"); sys.exit(0)
a.preRun(); chans_ref = orc.pre_run(os_, ref)
t = m.TrackingResult(a, device=0); ctx = m.engine.get_context(s, 0); r = ctx.upload(rec)
t.track(m.DeviceFile(r))
print("kernel", ctx.timing()["track_kernel"], "members", ctx.timing()["track_members"])
want = orc.stack_series(orc.track(os_, chans_ref, rec))
print("absoluteSample equal", np.array_equal(t.series[:, 0], want[:, 0]), "err", T._trk_err(t.series, want))
for c in range(want.shape[0]):
    bad = np.nonzero(t.series[c, 0] != want[c, 0])[0]
    print("ch", c, "first boundary diff", bad[:3], "max |dI_P|", np.max(np.abs(t.series[c, 3] - want[c, 3])))
    if bad.size:
        k = bad[0]
        print("  gpu", t.series[c, 0, max(0,k-2):k+3], "oracle", want[c, 0, max(0,k-2):k+3])
        print("  codeFreq gpu", t.series[c, 1, max(0,k-3):k+1], "oracle", want[c, 1, max(0,k-3):k+1])
