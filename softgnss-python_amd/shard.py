"""Multi-GPU sharding (one process per GPU).

Acquisition shards PRN indices, tracking shards channels; both are embarrassingly parallel.
The only exchange on the path is the acquisition peak gather: every rank contributes
(carrFreq, codePhase, peakMetric, freqBin, fineIdx) of its PRNs - 32 bytes per PRN - through
one ncclAllGather issued by libsgx (RCCL over xGMI).  A host gather - over the ranks' socket
rendezvous (rendezvous.HostGroup) or any object with torch.distributed's all_gather_object, e.g. the
gloo backend in the CPU tests - is the flagged fallback.
"""
import numpy as np

PEAK_DTYPE = np.dtype([("prn0", "<i4"), ("freqBin", "<i4"), ("carrFreq", "<f8"), ("codePhase", "<f8"),
                       ("peakMetric", "<f8"), ("fineIdx", "<i4"), ("valid", "<i4")])   # 40 bytes


def plan_shards(n_items, world):
    """Contiguous, balanced partition of range(n_items) over `world` ranks -> list of ranges."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(int(n_items), int(world))
    out = []
    lo = 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def pack_peaks(prn_indices, res, slots):
    """Fixed-size (slots) peak records of one rank; unused slots have valid = 0."""
    buf = np.zeros(slots, dtype=PEAK_DTYPE)
    for j, p in enumerate(prn_indices):
        buf[j] = (p, res["freqBin"][j], res["carrFreq"][j], res["codePhase"][j], res["peakMetric"][j],
                  res["fineIdx"][j], 1)
    return buf


def merge_peaks(gathered):
    """[world, slots] peak records -> the reference's three 32-entry result arrays + internals."""
    carr = np.zeros(32)
    cph = np.zeros(32)
    met = np.zeros(32)
    fb = np.full(32, -1, dtype=np.int64)
    fi = np.full(32, -1, dtype=np.int64)
    for rec in np.asarray(gathered).reshape(-1):
        if rec["valid"]:
            p = int(rec["prn0"])
            carr[p], cph[p], met[p] = rec["carrFreq"], rec["codePhase"], rec["peakMetric"]
            fb[p], fi[p] = rec["freqBin"], rec["fineIdx"]
    return dict(carrFreq=carr, codePhase=cph, peakMetric=met, freqBin=fb, fineIdx=fi)


class HostGather(object):
    """Peak gather through the host: `group` is a rendezvous.HostGroup, or anything else that offers
    get_world_size() and all_gather_object(out_list, obj) (torch.distributed with a CPU backend does)."""

    def __init__(self, group):
        self.group = group
        self.name = getattr(group, "name", "host")

    def allgather(self, buf):
        world = self.group.get_world_size()
        recv = [None] * world
        self.group.all_gather_object(recv, np.ascontiguousarray(buf).view(np.uint8).tobytes())
        return np.stack([np.frombuffer(r, dtype=np.uint8) for r in recv]).view(PEAK_DTYPE).reshape(world, -1)


class RcclGather(object):
    """Peak gather through libsgx's RCCL communicator (ncclAllGather over xGMI)."""
    name = "rccl"

    def __init__(self, comm):
        self.comm = comm

    def native(self):
        """The communicator sgx_acquire_sharded gathers with, on the device and on the search's own stream."""
        return self.comm

    def allgather(self, buf):
        out = self.comm.allgather(np.ascontiguousarray(buf).view(np.uint8))
        return out.view(PEAK_DTYPE).reshape(self.comm.n_ranks, -1)


class LocalGather(object):
    """world == 1 - or ONE rank's shard of a larger world run alone (what a rank of an N-GPU run executes, without the
    collective): only its own PRNs come back."""
    name = "local"

    def native(self):
        return None      # sgx_acquire_sharded without a communicator

    def allgather(self, buf):
        return np.asarray(buf).reshape(1, -1)


def acquire_sharded(acq, long_signal, rank, world, gather, n_prn=None, n_blocks=2, noncoh=False):
    """AcquisitionResult.acquire with the PRN search sharded over `world` ranks.

    Every rank searches its contiguous share of PRN indices on its own GPU, the peaks are
    all-gathered, and every rank ends with the same 32-entry result arrays as a single-GPU call.
    """
    settings = acq.settings
    if n_prn is None:
        n_prn = len(settings.acqSatelliteList)
    mine = list(plan_shards(n_prn, world)[rank])
    slots = -(-n_prn // world)
    if world == 1 and not isinstance(gather, RcclGather):
        # nothing to gather: the single-GPU call itself (a deferred AcquisitionResult stays queued)
        acq.acquire(long_signal, n_blocks=n_blocks, noncoh=noncoh, prn_indices=mine)
        return acq
    native = getattr(gather, "native", None)
    if native is not None and hasattr(long_signal, "record") and n_prn <= 32:
        # ONE library call: the search queued, the peaks packed on the device, one ncclAllGather on the same stream (or none:
        # a shard run alone), one look, the merge in C (sgx_acquire_sharded) - no Python between the kernels and the gather
        from . import engine
        ctx = engine.get_context(settings, acq._device)
        acq._fill32(ctx.acquire_sharded(native(), rank, world, long_signal.record, long_signal.offset, long_signal.length,
                                        n_prn_total=n_prn, n_blocks=n_blocks, noncoh=noncoh))
        return acq
    if mine:
        acq.acquire(long_signal, n_blocks=n_blocks, noncoh=noncoh, prn_indices=mine)
        res = dict(carrFreq=acq.carrFreq[mine], codePhase=acq.codePhase[mine], peakMetric=acq.peakMetric[mine],
                   freqBin=acq.internals["freqBin"][mine], fineIdx=acq.internals["fineIdx"][mine])
    else:
        res = dict(carrFreq=[], codePhase=[], peakMetric=[], freqBin=[], fineIdx=[])
    merged = merge_peaks(gather.allgather(pack_peaks(mine, res, slots)))
    acq.internals = dict(freqBin=merged["freqBin"], fineIdx=merged["fineIdx"])
    acq.results = np.rec.fromarrays([merged["carrFreq"], merged["codePhase"], merged["peakMetric"]],
                                    names='carrFreq,codePhase,peakMetric')
    return acq
