"""AcquisitionResult with the reference's interface (reference acquisition.py:6-336).

acquire() hands the 11 ms record to libsgx.so (sgx_acquire): carrier mix, batched 38192-point
FFT correlation against the C/A replicas over the Doppler grid, peak / second-peak test and the
2^22-point fine-frequency FFT all run in HIP kernels.  preRun() and showChannelStatus() are the
host glue between acquisition and tracking and stay in Python.
"""
from __future__ import print_function

import numpy as np

from . import engine
from .initialize import Result
from .record import DeviceSignal


class AcquisitionResult(Result):
    def __init__(self, settings, verbose=False, device=None, deferred=False):
        """deferred (extension; the reference and the default are eager): acquire() of a DeviceSignal only QUEUES the
        search and preRun() only notes that it was asked for; TrackingResult.track() of a DeviceFile then runs preRun on
        the device and the tracking kernel behind it and the host waits ONCE for all three (include/sgx.h,
        sgx_acquire_begin / sgx_track_chained).  Anything that looks at .results / .channels / .peakMetric ... earlier
        simply waits for the search then.  Results are the eager ones, bit for bit; the reference's IndexError
        (acquisition.py:152-162) is raised by whichever call looks first instead of by acquire()."""
        Result.__init__(self, settings)
        self._verbose = verbose
        self._device = device
        self._deferred = bool(deferred)
        self._pending = None           # (context, PRN indices) of a queued search nobody has looked at
        self._prerun_pending = False   # preRun() was called while the search was still queued
        self._merged = None            # arrays of a sharded search (sgx_acquire_sharded) not yet packed into .results
        self._internals = None

    @property
    def internals(self):
        """frequencyBinIndex / fftMaxIndex per PRN, for parity checks"""
        self._materialize()
        return self._internals

    @internals.setter
    def internals(self, value):
        self._internals = value

    def _materialize(self):
        merged = self._merged
        if merged is not None:
            self._merged = None
            self._results = np.rec.fromarrays([merged["carrFreq"], merged["codePhase"], merged["peakMetric"]],
                                              names='carrFreq,codePhase,peakMetric')
        if self._pending is not None:
            ctx, prn_indices, token = self._pending
            self._pending = None
            if getattr(ctx, "_acq_token", None) != token:
                raise RuntimeError("this deferred acquisition was superseded by a later one on the same context before "
                                   "anybody looked at it (one search may be pending per context)")
            self._fill(prn_indices, ctx.acquire_end(len(prn_indices)))
        if self._prerun_pending:
            self._prerun_pending = False
            self._prerun_host()

    @property
    def peakMetric(self):
        return self.results.peakMetric

    @property
    def carrFreq(self):
        return self.results.carrFreq

    @property
    def codePhase(self):
        return self.results.codePhase

    def acquire(self, longSignal, n_blocks=2, noncoh=False, prn_indices=None):
        """Cold-start acquisition (reference acquisition.py:27-204).

        longSignal  1-D int8 samples (11 ms: the fine search needs codePhase + 10 ms), or a
                    DeviceSignal window of a record already resident in HBM.
        The reference searches PRN indices range(len(acqSatelliteList)) - the list's VALUES
        are ignored - and so does this method unless prn_indices is given.
        n_blocks / noncoh are extensions (reference behaviour: 2, False).
        Raises IndexError exactly where the reference does (coarse code phase == 37 samples, Q5).
        """
        settings = self._settings
        ctx = engine.get_context(settings, self._device)
        if prn_indices is None:
            prn_indices = range(len(settings.acqSatelliteList))
        prn_indices = [int(p) for p in prn_indices]
        own = None
        f64 = None
        if isinstance(longSignal, DeviceSignal):
            rec, off, n = longSignal.record, longSignal.offset, longSignal.length
        else:
            arr = np.asarray(longSignal)
            if arr.ndim != 1:
                raise ValueError("longSignal must be one-dimensional")
            if arr.dtype == np.int8:
                own = rec = ctx.upload(arr)
                off, n = 0, arr.size
            else:
                # the reference works on whatever real dtype it is handed (acquisition.py:55-59): fp64 copy in HBM
                if not np.isrealobj(arr):
                    raise TypeError("longSignal must be real-valued")
                f64 = arr.astype(np.float64)
        self._prerun_pending = False
        self._merged = None
        if self._deferred and f64 is None and own is None and not self._verbose:
            token = ctx.acquire_begin(rec, off, n, prn_indices, n_blocks=n_blocks, noncoh=noncoh)
            self._pending = (ctx, prn_indices, token)
            self._results = None
            self._channels = None
            return
        self._pending = None
        if self._verbose:
            print('(')
        try:
            if f64 is not None:
                r = ctx.acquire_f64(f64, prn_indices, n_blocks=n_blocks, noncoh=noncoh)
            else:
                r = ctx.acquire(rec, off, n, prn_indices, n_blocks=n_blocks, noncoh=noncoh)
        finally:
            if own is not None:
                own.free()
        self._fill(prn_indices, r)
        return

    def _fill32(self, r):
        """The merged 32-entry arrays of a sharded search (sgx_acquire_sharded) as this object's results."""
        self._pending = None
        self._prerun_pending = False
        self._internals = dict(freqBin=r["freqBin"], fineIdx=r["fineIdx"])
        self._results = None
        self._merged = r               # the record array (acquisition.py:201-203) is packed on first access

    def _fill(self, prn_indices, r):
        """The reference's three 32-entry result arrays (acquisition.py:201-203) from the library's per-PRN outputs."""
        carrFreq = np.zeros(32)
        codePhase_ = np.zeros(32)
        peakMetric = np.zeros(32)
        freqBin = np.full(32, -1, dtype=np.int64)
        fineIdx = np.full(32, -1, dtype=np.int64)
        idx = np.asarray(prn_indices, dtype=np.int64)
        carrFreq[idx] = r["carrFreq"]
        codePhase_[idx] = r["codePhase"]
        peakMetric[idx] = r["peakMetric"]
        freqBin[idx] = r["freqBin"]
        fineIdx[idx] = r["fineIdx"]
        if self._verbose:
            for p in prn_indices:
                print('%02d ' % (p + 1) if carrFreq[p] > 0 else '. ')
            print(')\n')
        self._internals = dict(freqBin=freqBin, fineIdx=fineIdx)
        self._results = np.rec.fromarrays([carrFreq, codePhase_, peakMetric],
                                          names='carrFreq,codePhase,peakMetric')
        return

    def plot(self):
        """Bar chart of the acquisition metric per PRN, acquired signals highlighted (reference
        acquisition.py:206-256).  Needs matplotlib; prints a notice and returns without it."""
        from .initialize import _pyplot
        plt = _pyplot("AcquisitionResult.plot")
        if plt is None:
            return
        assert isinstance(self.results, np.recarray)
        plt.figure(101)
        plt.clf()
        prn = np.arange(1, len(self.peakMetric) + 1)
        plt.bar(prn, self.peakMetric)
        found = self.carrFreq > 0
        plt.bar(prn[found], self.peakMetric[found], color=(0, 0.8, 0))
        plt.title('Acquisition results')
        plt.xlabel('PRN number (no bar - SV is not in the acquisition list)')
        plt.ylabel('Acquisition Metric')
        plt.legend(['Not acquired signals', 'Acquired signals'])

    def preRun(self):
        """Channel table from the acquisition results (reference acquisition.py:259-306): stable
        descending sort on peakMetric, first min(numberOfChannels, #detected) become channels."""
        if self._pending is not None:
            # deferred: the same table is made on the device when TrackingResult.track() queues the tracking kernel
            # (csrc/sgx_acq.hip: acq_prerun_kernel), or here on the host by the first look at .channels
            self._prerun_pending = True
            self._channels = None
            return
        self._prerun_host()

    def _channels_from_table(self, prn, freq, cph, n_active):
        """The channel table as the device-side preRun left it (sgx_track_chained)."""
        nch = len(prn)
        status = ['T' if i < n_active else '-' for i in range(nch)]
        self._prerun_pending = False
        self._channels = np.rec.fromarrays([np.asarray(prn, dtype='int64'), np.asarray(freq, dtype=float),
                                            np.asarray(cph, dtype=float), status],
                                           names='PRN,acquiredFreq,codePhase,status')

    def _prerun_host(self):
        assert isinstance(self.results, np.recarray)
        settings = self._settings
        nch = int(settings.numberOfChannels)
        PRN = np.zeros(nch, dtype='int64')
        acquiredFreq = np.zeros(nch)
        codePhase = np.zeros(nch)
        status = ['-' for _ in range(nch)]
        order = sorted(enumerate(self.peakMetric), key=lambda x: x[-1], reverse=True)
        for ii in range(min(nch, int(np.sum(self.carrFreq > 0)))):
            idx = order[ii][0]
            PRN[ii] = idx + 1
            acquiredFreq[ii] = self.carrFreq[idx]
            codePhase[ii] = self.codePhase[idx]
            status[ii] = 'T'
        self._channels = np.rec.fromarrays([PRN, acquiredFreq, codePhase, status],
                                           names='PRN,acquiredFreq,codePhase,status')
        return

    def showChannelStatus(self):
        """ASCII channel table (reference acquisition.py:308-336)."""
        channel = self._channels
        settings = self._settings
        assert isinstance(channel, np.recarray)
        bar = '*=========*=====*===============*===========*=============*========*'
        print('\n' + bar)
        print('| Channel | PRN |   Frequency   |  Doppler  | Code Offset | Status |')
        print(bar)
        for channelNr in range(settings.numberOfChannels):
            if channel[channelNr].status != '-':
                print('|      %2d | %3d |  %2.5e |   %5.0f   |    %6d   |     %1s  |' % (
                    channelNr, channel[channelNr].PRN, channel[channelNr].acquiredFreq,
                    channel[channelNr].acquiredFreq - settings.IF, channel[channelNr].codePhase,
                    channel[channelNr].status))
            else:
                print('|      %2d | --- |  ------------ |   -----   |    ------   |   Off  |' % channelNr)
        print(bar + '\n')
