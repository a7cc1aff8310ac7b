"""Two REAL ranks through the real package on the one GPU the driver has: two fresh
child processes, RANK 0 / 1, both on device 0, walk rendezvous.HostGroup -> sgx_comm_create (RCCL refuses two ranks on one
device: the refusal must be flagged and every rank must fall back to the host gather) -> shard.acquire_sharded for BASELINE
configs[3] -> each rank tracks its half of the 8 channels.  The merged acquisition equals the single-rank one bit for bit on
every rank, and the ranks' tracking results, put together, equal the 8-channel run of this process bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_gpu_shard_acquisition_and_tracking(tmp_path):
    ms, world = 2000, 2
    port, token = _free_port(), os.urandom(16).hex()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), SGX_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SGX_RDV_TOKEN=token, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("SGX_LIB", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_child.py"), str(tmp_path), str(ms)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish in 600 s")
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][1].decode(errors="replace")[-3000:])
    reps = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]
    for rep in reps:
        # two ranks on one device: RCCL must refuse, the refusal must be visible, and EVERY rank must be on the host gather
        assert rep["rccl_error"], rep
        assert rep["peak_gather"] != "rccl" and len(set(r2["peak_gather"] for r2 in reps)) == 1
        assert rep["acq_equal"], rep                    # the merged result of the sharded search = the single-rank search
        assert rep["done"] == [ms] * len(rep["channels"])
        assert rep["track_kernel"] == 5 and rep["track_members"] == 20      # the headline kernel, the headline layout
    assert reps[0]["detected"] == reps[1]["detected"] and len(reps[0]["detected"]) == 8
    assert sorted(reps[0]["channels"] + reps[1]["channels"]) == list(range(8))
    # the 8-channel run, in this process
    m = pkg()
    s = m.Settings()
    s.msToProcess = float(ms)
    s.numberOfChannels = 8
    n = s.samplesPerCode
    ctx = m.engine.get_context(s, 0)
    rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
    a = m.AcquisitionResult(s, device=0)
    a.acquire(m.DeviceSignal(rec, 0, 20 * n), n_blocks=10, noncoh=True)
    a.preRun()
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    want, done = ctx.track(rec, chans, ms)
    rec.free()
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "series_rank%d.npy" % r)) for r in range(world)])
    assert np.all(done == ms)
    assert np.array_equal(got, np.asarray(want))        # bit for bit: boundaries AND correlator series
