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
    def __init__(self, settings, verbose=False, device=None):
        Result.__init__(self, settings)
        self._verbose = verbose
        self._device = device
        self.internals = None      # frequencyBinIndex / fftMaxIndex per PRN, for parity checks

    @property
    def peakMetric(self):
        assert isinstance(self._results, np.recarray)
        return self._results.peakMetric

    @property
    def carrFreq(self):
        assert isinstance(self._results, np.recarray)
        return self._results.carrFreq

    @property
    def codePhase(self):
        assert isinstance(self._results, np.recarray)
        return self._results.codePhase

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
        carrFreq = np.zeros(32)
        codePhase_ = np.zeros(32)
        peakMetric = np.zeros(32)
        freqBin = np.full(32, -1, dtype=np.int64)
        fineIdx = np.full(32, -1, dtype=np.int64)
        for j, p in enumerate(prn_indices):
            carrFreq[p] = r["carrFreq"][j]
            codePhase_[p] = r["codePhase"][j]
            peakMetric[p] = r["peakMetric"][j]
            freqBin[p] = r["freqBin"][j]
            fineIdx[p] = r["fineIdx"][j]
            if self._verbose:
                print('%02d ' % (p + 1) if carrFreq[p] > 0 else '. ')
        if self._verbose:
            print(')\n')
        self.internals = dict(freqBin=freqBin, fineIdx=fineIdx)
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
        assert isinstance(self._results, np.recarray)
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
        assert isinstance(self._results, np.recarray)
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
