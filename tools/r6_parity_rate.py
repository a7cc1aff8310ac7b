"""(round 6) How often is config 3 bit-identical in its block boundaries, and what is the first divergence when it is not?
BASELINE configs[2] at full size (8 channels x 37 000 ms) against the numpy oracle on the default scene and N random scenes.
For every channel whose absoluteSample series differs: the first block whose correlator sums differ by more than 1e-8 of
their scale (the "blip": one sample on the other side of a chip boundary; everything before agrees to ~3e-11), the size of
the blip in units of that block's samples, and - from the ORACLE's own code phase and rate at that block, recomputed with the
reference's arithmetic (tracking.py:148-190) - the distance in chips from the nearest sample of any arm to a chip boundary.
The claim of DESIGN.md section 2 is that this distance is ~1e-11 chips or less whenever the boundaries differ.
GPU box:  python3 tools/r6_parity_rate.py 16 > gpurun_out/r06_full_parity.jsonl
"""
import importlib, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle_helpers import oracle_channel
from full_parity import random_scene

FS = 38.192e6


def rem_at(x, k_end):
    """remCodePhase at the start of block k_end and the code rate used in it, from a channel's recorded series x[13, ms]
    (absoluteSample, codeFreq), with the reference's own arithmetic."""
    pos = np.concatenate([[x[0, 0] - 38192.0], x[0]])
    rem, cf = 0.0, 1.023e6
    for k in range(k_end):
        blk = int(pos[k + 1] - pos[k])
        step = cf / FS
        stp = ((blk * step + rem) - rem) / blk
        rem = ((blk - 1) * stp + rem) + step - 1023.0
        cf = x[1, k]
    return rem, cf


def nearest_boundary(rem, cf):
    """min over samples and arms of the distance from a sample's code phase to an integer (a chip boundary of ceil)."""
    step = cf / FS
    blk = int(np.ceil((1023.0 - rem) / step))
    best = (1.0, None, None)
    for arm, off in (("E", -0.5), ("L", 0.5), ("P", 0.0)):
        t = np.linspace(rem + off, blk * step + rem + off, blk, endpoint=False)
        d = np.abs(t - np.round(t))
        i = int(np.argmin(d))
        if d[i] < best[0]:
            best = (float(d[i]), arm, i)
    return best


def main():
    n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    only = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None   # (just these seeds, no default scene)
    ms = 37000
    m = importlib.import_module("softgnss-python_amd")
    s = m.Settings()
    s.msToProcess = float(ms)
    ctx = m.engine.get_context(s, 0)
    n = s.samplesPerCode
    for seed in (only if only else [None] + list(range(seed0, seed0 + n_random))):
        scene = m.synth.Scene.default() if seed is None else random_scene(m, seed)
        rec = ctx.synth(scene, m.synth.record_length(n, ms))
        a = m.AcquisitionResult(s, device=0)
        a.acquire(m.DeviceSignal(rec, 0, 11 * n))
        a.preRun()
        chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
        got, done = ctx.track(rec, chans, ms)
        got = np.array(got)
        host = rec.download()
        t0 = time.time()
        with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            want = np.stack(list(ex.map(oracle_channel, [(host, p, f, c, ms) for p, f, c in chans])))
        scale = np.sqrt(np.mean(want[:, 3] ** 2 + want[:, 7] ** 2, axis=1))
        out = dict(scene="default" if seed is None else "random seed %d" % seed, channels=len(chans), blocks=len(chans) * ms,
                   absoluteSample_identical=bool(np.array_equal(got[:, 0], want[:, 0])), kernel=int(ctx.timing()["track_kernel"]),
                   oracle_seconds=round(time.time() - t0, 1), divergences=[])
        errs = []
        for ch in range(len(chans)):
            rel = np.max(np.abs(got[ch, 3:9] - want[ch, 3:9]), axis=0) / max(1.0, scale[ch])     # per block
            bad = np.nonzero(got[ch, 0] != want[ch, 0])[0]
            k_end = int(bad[0]) if bad.size else ms
            errs.append(float(rel[:k_end].max()) if k_end else 0.0)
            blips = np.nonzero(rel > 1e-8)[0]
            if bad.size or blips.size:
                j = int(blips[0]) if blips.size else None
                d = dict(channel=ch, prn=chans[ch][0], first_boundary_difference_block=int(bad[0]) if bad.size else None,
                         first_blip_block=j)
                if j is not None:
                    rem, cf = rem_at(want[ch], j)
                    dist, arm, i = nearest_boundary(rem, cf)
                    pos0 = int(want[ch, 0, j - 1]) if j else int(want[ch, 0, 0] - 38192)
                    x = int(host[pos0 + i]) if i is not None else 0
                    after = {str(o): float(rel[j + o]) for o in (1, 2, 5, 10, 50, 100, 500, 1000, 5000) if j + o < ms}
                    d.update(rel_err_at_blip=float(rel[j]), rel_err_blocks_after=after, rel_err_max_last_1000=float(rel[-1000:].max()),
                             code_freq_diff_after_Hz={str(o): float(abs(got[ch, 1, j + o] - want[ch, 1, j + o])) for o in (1, 100, 1000) if j + o < ms},
                             blip_abs=float(np.max(np.abs(got[ch, 3:9, j] - want[ch, 3:9, j]))), sample_value=x,
                             nearest_sample_to_a_chip_boundary_chips=dist, arm=arm, sample_index_in_block=i,
                             max_rel_err_before_blip=float(rel[:j].max()) if j else 0.0)
                out["divergences"].append(d)
        out["max_rel_err_IQ_while_identical"] = max(errs)
        print(json.dumps(out))
        sys.stdout.flush()
        rec.free()


if __name__ == "__main__":
    main()
