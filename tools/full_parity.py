"""BASELINE config 3 in full against the oracle: 8 channels x 37 000 ms of the default scene, GPU vs the numpy
restatement of the reference (about four minutes of host time on the GPU box).  Prints one JSON line.
Usage: python tools/full_parity.py [ms [scene-seed]]"""
import importlib, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_helpers import oracle_channel   # (the tests' helper; tests never import from tools/)


def random_scene(m, seed):
    """Eight satellites with random PRNs, Dopplers, code phases and amplitudes (scene seed != the default's)."""
    rng = np.random.default_rng(seed)
    prns = sorted(rng.choice(np.arange(1, 33), size=8, replace=False).tolist())
    return m.synth.Scene.make(0x50AC0000 + seed, 38192000.0, 9548000.0, prns,
                              [float(rng.uniform(-6500, 6500)) for _ in prns],
                              [int(rng.integers(0, 38192)) for _ in prns], [int(rng.integers(5, 10)) for _ in prns])


def main():
    ms = int(sys.argv[1]) if len(sys.argv) > 1 else 37000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else None
    m = importlib.import_module("softgnss-python_amd")
    s = m.Settings()
    s.msToProcess = float(ms)
    ctx = m.engine.get_context(s, 0)
    n = s.samplesPerCode
    scene = m.synth.Scene.default() if seed is None else random_scene(m, seed)
    rec = ctx.synth(scene, m.synth.record_length(n, ms))
    a = m.AcquisitionResult(s, device=0)
    a.acquire(m.DeviceSignal(rec, 0, 11 * n))
    a.preRun()
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    got, done = ctx.track(rec, chans, ms)
    host = rec.download()
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        want = list(ex.map(oracle_channel, [(host, p, f, c, ms) for p, f, c in chans]))
    want = np.stack(want)
    scale = np.sqrt(np.mean(want[:, 3] ** 2 + want[:, 7] ** 2, axis=1))
    err = np.max(np.abs(got[:, 3:9] - want[:, 3:9]), axis=(1, 2)) / np.maximum(1.0, scale)
    out = dict(ms=ms, scene="default" if seed is None else "random seed %d" % seed, channels=len(chans),
               blocks=int(len(chans) * ms),
               absoluteSample_identical=bool(np.array_equal(got[:, 0], want[:, 0])),
               max_rel_err_IQ=float(err.max()), max_abs_err_codeFreq_Hz=float(np.max(np.abs(got[:, 1] - want[:, 1]))),
               max_abs_err_carrFreq_Hz=float(np.max(np.abs(got[:, 2] - want[:, 2]))),
               oracle_seconds=round(time.time() - t0, 1), kernel_ms=ctx.timing()["track_ms"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
