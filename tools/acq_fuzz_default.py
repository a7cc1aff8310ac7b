"""(GPU box) Acquisition at the DEFAULT front end (38 192 samples per code: the four-step kernels and the fused fine search)
on random scenes against the oracle: codePhase, carrFreq and the bins exactly, peakMetric to 1e-9; 2 x 1 ms and the
10 ms non-coherent extension.  Usage: python tools/acq_fuzz_default.py <first seed> <last seed + 1>"""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
m = T.pkg()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    rng = random.Random(0xACF0 + seed)
    n_sat = rng.randrange(0, 12)
    prns = rng.sample(range(1, 33), n_sat)
    s = m.Settings(); so = T.orc.OracleSettings()
    n = s.samplesPerCode
    sc = m.synth.Scene.make(0xACF00000 + seed, s.samplingFreq, s.IF, prns, [rng.uniform(-6900, 6900) for _ in prns],
                            [rng.randrange(0, n) for _ in prns], [rng.choice([3, 4, 5, 6, 8, 10]) for _ in prns])
    x = m.synth.generate(sc, 21 * n)
    try:
        a = m.AcquisitionResult(s, device=0); a.acquire(x[:11 * n])
        w = T.orc.acquire(so, x[:11 * n])
        ok = (np.array_equal(a.codePhase, w["codePhase"]) and np.array_equal(a.carrFreq, w["carrFreq"]) and
              np.allclose(a.peakMetric, w["peakMetric"], rtol=1e-9, atol=0))
        print("seed %d: %d satellites, %d detected, %s" % (seed, n_sat, int((a.carrFreq > 0).sum()), "ok" if ok else "MISMATCH"))
        if not ok:
            bad.append(seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", repr(e)[:300])
print("seeds %d..%d: %d failures %s" % (lo, hi - 1, len(bad), bad))
