"""One rank of the two-process test in tests/test_two_ranks_gpu.py (started with RANK / WORLD_SIZE / MASTER_* / SGX_RDV_TOKEN in
the environment, both ranks on device 0).  Walks exactly what `bench.py --gpus N` walks per rank - rendezvous.HostGroup ->
the RCCL communicator (refused when two ranks share a device: flagged, everybody falls back to the host gather) ->
shard.acquire_sharded for BASELINE configs[3] -> sharded tracking - and leaves its results in the directory given."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir, ms = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pkg = importlib.import_module("softgnss-python_amd")
shard = importlib.import_module("softgnss-python_amd.shard")
rendezvous = importlib.import_module("softgnss-python_amd.rendezvous")

s = pkg.Settings()
s.msToProcess = float(ms)
s.numberOfChannels = 8
n = s.samplesPerCode
group = rendezvous.HostGroup(rank, world)
ctx = pkg.engine.get_context(s, 0)
report = {"rank": rank, "rccl_error": None}
try:
    uid = group.broadcast(pkg._native.Comm.unique_id() if rank == 0 else None)
    gather = shard.RcclGather(pkg._native.Comm(ctx, world, rank, uid))
    gather.allgather(shard.pack_peaks([], dict(), 1))
except Exception as e:   # noqa: BLE001 - the refusal is what the test looks for
    report["rccl_error"] = str(e)
    gather = shard.HostGather(group)
names = group.gather(gather.name)
if len(set(names)) != 1:
    gather = shard.HostGather(group)
report["transports_tried"] = names
report["peak_gather"] = gather.name

rec = ctx.synth(pkg.synth.Scene.default(), pkg.synth.record_length(n, ms))
sig4 = pkg.DeviceSignal(rec, 0, 20 * n)
a = pkg.AcquisitionResult(s, device=0)
shard.acquire_sharded(a, sig4, rank, world, gather, n_blocks=10, noncoh=True)          # BASELINE configs[3], sharded
one = pkg.AcquisitionResult(s, device=0)
one.acquire(sig4, n_blocks=10, noncoh=True)                                            # ... and on this rank alone
report["acq_equal"] = bool(np.array_equal(a.carrFreq, one.carrFreq) and np.array_equal(a.codePhase, one.codePhase) and
                           np.array_equal(a.peakMetric, one.peakMetric) and
                           np.array_equal(a.internals["freqBin"], one.internals["freqBin"]) and
                           np.array_equal(a.internals["fineIdx"], one.internals["fineIdx"]))
report["detected"] = [int(i) for i in np.nonzero(a.carrFreq)[0]]
a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
mine = list(shard.plan_shards(len(chans), world)[rank])
# (one GPU under both ranks: the cooperative tracking launches take turns - each wants more than half the CUs)
for turn in range(world):
    if turn == rank:
        series, done = ctx.track(rec, [chans[i] for i in mine], ms)
        report["track_kernel"] = int(ctx.timing()["track_kernel"])
        report["track_members"] = int(ctx.timing()["track_members"])
    group.barrier()
np.save(os.path.join(out_dir, "series_rank%d.npy" % rank), np.asarray(series))
report["channels"] = mine
report["done"] = [int(d) for d in done]
with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
    json.dump(report, f)
rec.free()
group.close()
