"""Stand-in for the `softgnss-python_amd` package in the CPU test of bench.py's N > 1 plumbing (SGX_BENCH_PKG=bench_stub):
no GPU, no library.  It answers the calls bench.py makes with the reference-made acquisition golden and a fabricated
tracking series, so that what the test exercises is bench.py itself - self-launch, rendezvous, PRN sharding, the peak
gather's fallback, the max over ranks and the JSON line.  Test infrastructure only."""
import os
import types

import numpy as np

_G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "acq_default.npz"))


class Settings(object):
    def __init__(self):
        self.msToProcess = 37000.0
        self.numberOfChannels = 8
        self.acqSearchBand = 14.0
        self.acqSatelliteList = list(range(1, 33))
        self.samplesPerCode = 38192


class _Record(object):
    def __init__(self, n):
        self.n = n

    def free(self):
        pass


class _Ctx(object):
    def __init__(self):
        self._t = dict(acquire_ms=1.0, acq_coarse_ms=float('nan'), acq_fine_ms=float('nan'), track_ms=50.0, synth_ms=1.0)

    def synth(self, scene, n):
        return _Record(n)

    def timing(self):
        return dict(self._t)

    def sync(self):
        pass

    def stream_rates(self, nbytes, reps):
        return 4000.0, 2000.0


_CTX = _Ctx()


def _get_context(s, device=None):
    if os.environ.get("SGX_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit("stand-in: rank %s has no device (test hook)" % os.environ.get("RANK"))
    return _CTX


engine = types.SimpleNamespace(get_context=_get_context)
synth = types.SimpleNamespace(Scene=types.SimpleNamespace(default=staticmethod(lambda: None)),
                              record_length=lambda n_code, ms: (ms + 1) * (n_code + 1) + n_code)


def DeviceSignal(rec, offset, n):
    return (rec, offset, n)


def DeviceFile(rec):
    return rec


class AcquisitionResult(object):
    def __init__(self, settings, device=0, deferred=False):
        self.settings = settings
        self._device = device
        self.internals = None
        self.results = None
        self.searched = []

    def acquire(self, signal, n_blocks=2, noncoh=False, prn_indices=None):
        self.searched = list(prn_indices) if prn_indices is not None else list(range(32))
        self.carrFreq = _G["carrFreq"].copy()
        self.codePhase = _G["codePhase"].copy()
        self.peakMetric = _G["peakMetric"].copy()
        self.internals = dict(freqBin=_G["freqBin"].copy(), fineIdx=_G["fineIdx"].copy())
        if prn_indices is None or len(self.searched) == 32:   # (the whole search: the real class fills in its results)
            self.results = np.rec.fromarrays([self.carrFreq, self.codePhase, self.peakMetric],
                                             names="carrFreq,codePhase,peakMetric")

    def __getattr__(self, name):
        if name in ("carrFreq", "codePhase", "peakMetric") and self.__dict__.get("results") is not None:
            return self.results[name]
        raise AttributeError(name)

    def preRun(self):
        r = self.results
        order = np.argsort(-r["peakMetric"], kind="stable")[:self.settings.numberOfChannels]
        rows = [(int(p) + 1, float(r["carrFreq"][p]), float(r["codePhase"][p])) if r["carrFreq"][p] > 0 else (0, 0.0, 0.0)
                for p in order]
        self.channels = np.rec.fromrecords(rows, names="PRN,acquiredFreq,codePhase")


class TrackingResult(object):
    chained = False

    def __init__(self, acq, device=0):
        self.acq = acq
        self.kernel_ms = 50.0

    def track(self, rec):
        ms = int(self.acq.settings.msToProcess)
        act = [c for c in self.acq.channels if c.PRN != 0]
        self.series = np.zeros((len(act), 13, ms))
        for i, c in enumerate(act):
            self.series[i, 0] = c.codePhase + 38192.0 * np.arange(1, ms + 1)


class _NoRccl(object):
    @staticmethod
    def unique_id():
        return b"\0" * 128

    def __init__(self, *a):
        raise RuntimeError("the stand-in package has no RCCL")


_native = types.SimpleNamespace(device_count=lambda: int(os.environ.get("SGX_STUB_DEVICES", "2")),
                                pinned_empty=lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype), Comm=_NoRccl)
