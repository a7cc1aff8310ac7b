"""Run the randomised front-end / scene parity check of tests/test_gpu_parity.py over many more seeds.
Usage (GPU box): python tools/fuzz_parity.py 100 200"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    try:
        T.test_random_front_ends_and_scenes_against_oracle(seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", repr(e)[:300])
        traceback.print_exc(limit=2)
print("seeds %d..%d: %d failures %s" % (lo, hi - 1, len(bad), bad))


def edge_case(fs, phase, seed):
    """One strong satellite whose code starts `phase` samples into the record (edges of the code period)."""
    import numpy as np
    m = T.pkg()
    s = m.Settings(); os_ = T.orc.OracleSettings()
    for o in (s, os_):
        o.samplingFreq, o.IF, o.acqSatelliteList = fs, 0.25 * fs, [1]
    n = s.samplesPerCode
    sc = m.synth.Scene.make(0xED6E0000 + seed, fs, s.IF, [1], [1500], [phase % n], [10])
    x = m.synth.generate(sc, 11 * n)
    a = m.AcquisitionResult(s, device=0)
    try:
        want = T.orc.acquire(os_, x)
        werr = None
    except Exception as e:   # noqa: BLE001
        want, werr = None, type(e)
    try:
        a.acquire(x)
        gerr = None
    except Exception as e:   # noqa: BLE001
        gerr = type(e)
    if werr or gerr:
        return werr == gerr, "exceptions %s / %s" % (werr, gerr)
    ok = np.array_equal(a.codePhase, want["codePhase"]) and np.array_equal(a.carrFreq, want["carrFreq"]) and \
        np.allclose(a.peakMetric, want["peakMetric"], rtol=1e-9, atol=0)
    return ok, "codePhase %s / %s" % (a.codePhase[0], want["codePhase"][0])


if len(sys.argv) > 3 and sys.argv[3] == "edges":
    nbad = 0
    for fs in (38192000.0, 16367600.0, 5456000.0, 26000000.0):
        n = int(round(fs / 1000))
        spc = int(round(fs / 1.023e6))
        for phase in sorted(set(list(range(0, 4)) + list(range(spc - 3, spc + 4)) + list(range(n - spc - 3, n - spc + 4)) + [n - 2, n - 1])):
            ok, info = edge_case(fs, phase + 1, phase)
            nbad += (not ok)
            if not ok:
                print("fs %.4g start %d: MISMATCH %s" % (fs, phase, info))
    print("edge cases: %d mismatches" % nbad)
