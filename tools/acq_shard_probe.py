"""Per-GPU acquisition time as a function of the PRN shard size (what each rank of an N-GPU run executes),
reference semantics (2 x 1 ms) and the 10 ms non-coherent extension (BASELINE config 4).  GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
shard = importlib.import_module("softgnss-python_amd.shard")
s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 21 * n)
for n_blocks, noncoh, label in ((2, False, "2 x 1 ms (reference)"), (10, True, "10 ms non-coherent")):
    sig = m.DeviceSignal(rec, 0, (10 + n_blocks) * n if noncoh else 11 * n)
    base = None
    for world in (1, 2, 4, 8):
        mine = list(shard.plan_shards(32, world)[0])          # rank 0's share: PRNs 1.. (holds PRN 1, 3: detections)
        worst = 0.0
        for rank in range(world):
            prns = list(shard.plan_shards(32, world)[rank])
            a = m.AcquisitionResult(s, device=0)
            a.acquire(sig, n_blocks=n_blocks, noncoh=noncoh, prn_indices=prns)
            ts = []
            for _ in range(3):
                a.acquire(sig, n_blocks=n_blocks, noncoh=noncoh, prn_indices=prns)
                ts.append(ctx.timing()["acquire_ms"])
            worst = max(worst, min(ts))
        base = base or worst
        print("%-22s world %d: slowest rank %.3f ms  -> speed-up %.2fx" % (label, world, worst, base / worst))
