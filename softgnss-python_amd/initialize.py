"""Settings / Result / TruePosition with the reference's names (reference initialize.py:20-185).

The three helpers the hot path calls - generateCAcode, makeCaTable, calcLoopCoef - are answered
by libsgx.so's exact host routines (include/sgx.h), not by Python arithmetic.
"""
import ctypes as C
import datetime

import numpy as np

from . import _native


class Result(object):
    """Base of AcquisitionResult / TrackingResult (reference initialize.py:20-46)."""

    def __init__(self, settings):
        self._settings = settings
        self._results = None
        self._channels = None
        self._merged = None

    @property
    def settings(self):
        return self._settings

    def _materialize(self):
        """Hook of the subclasses: results that are still queued on the GPU, or not yet packed into record arrays, are
        brought in before anybody looks (the reference computes everything eagerly; so does this class by default)."""

    @property
    def channels(self):
        self._materialize()
        assert isinstance(self._channels, np.recarray)
        return self._channels

    @property
    def results(self):
        self._materialize()
        assert isinstance(self._results, np.recarray)
        return self._results

    @results.setter
    def results(self, records):
        assert isinstance(records, np.recarray)
        self._merged = None        # (AcquisitionResult: arrays of a sharded search not yet packed)
        self._results = records

    def plot(self):
        pass


def _pyplot(what):
    """matplotlib.pyplot, or None (with a one-line notice) where matplotlib is not installed: a script that calls
    plot() like the reference's postProcessing does keeps running."""
    try:
        import matplotlib.pyplot as plt
        return plt
    except ImportError:
        print("   (%s: matplotlib is not installed, nothing drawn)" % what)
        return None


class TruePosition(object):
    """E/N/U holder (reference initialize.py:49-77)."""

    def __init__(self):
        self.E = self.N = self.U = None


# attribute -> default, the receiver configuration of reference initialize.py:85-173 (same names, same values)
_DEFAULTS = (
    ("msToProcess", 37000.0), ("numberOfChannels", 8), ("skipNumberOfBytes", 0),
    ("fileName", 'GPSdata-DiscreteComponents-fs38_192-if9_55.bin'), ("dataType", 'int8'),
    ("IF", 9548000.0), ("samplingFreq", 38192000.0), ("codeFreqBasis", 1023000.0), ("codeLength", 1023),
    ("skipAcquisition", False), ("acqSearchBand", 14.0), ("acqThreshold", 2.5),
    ("dllDampingRatio", 0.7), ("dllNoiseBandwidth", 2.0), ("dllCorrelatorSpacing", 0.5),
    ("pllDampingRatio", 0.7), ("pllNoiseBandwidth", 25.0),
    ("navSolPeriod", 500.0), ("elevationMask", 10.0), ("useTropCorr", True), ("plotTracking", True),
)


class Settings(object):
    """Receiver configuration; attribute names and defaults of reference initialize.py:81-173."""

    c = property(lambda self: 299792458.0, doc="speed of light, m/s (read-only, initialize.py:170)")
    startOffset = property(lambda self: 68.802, doc="initial travel-time guess, ms (read-only, initialize.py:172)")

    def __init__(self):
        for name, value in _DEFAULTS:
            setattr(self, name, value)
        self.acqSatelliteList = range(1, 33)      # PRN indices 0..31 are searched (acquisition.py:103)
        self.truePosition = TruePosition()

    @property
    def samplesPerCode(self):
        n = C.c_int64(0)
        st = _native.settings_struct(self)
        _native.check(_native.lib().sgx_samples_per_code(C.byref(st), C.byref(n)))
        return int(n.value)

    def makeCaTable(self):
        """float64[32, samplesPerCode] sampled C/A codes (reference initialize.py:188-231)."""
        st = _native.settings_struct(self)
        out = np.empty((32, self.samplesPerCode))
        _native.check(_native.lib().sgx_make_ca_table(C.byref(st), out.ctypes.data_as(C.c_void_p)))
        return out

    def generateCAcode(self, prn):
        """float64[1023] of +-1 for PRN index 0..31 (reference initialize.py:234-302)."""
        assert prn in range(0, 32)
        out = np.empty(1023)
        _native.check(_native.lib().sgx_generate_ca_code(int(prn), out.ctypes.data_as(C.c_void_p)))
        return out

    @staticmethod
    def calcLoopCoef(LBW, zeta, k):
        """(tau1, tau2) of the loop filter (reference initialize.py:304-328)."""
        t1 = C.c_double(0)
        t2 = C.c_double(0)
        _native.check(_native.lib().sgx_calc_loop_coef(float(LBW), float(zeta), float(k), C.byref(t1),
                                                       C.byref(t2)))
        return t1.value, t2.value

    def probeData(self, fileNameStr=None, device=None):
        """Raw-data information of reference initialize.py:330-417 for the first 10 code periods of the record:
        time-domain samples, Welch power spectral density (16384-point periodic Hamming window, 1024 overlap) and
        the histogram.  The numbers come from sgx_probe_stats on the GPU; they are returned as a dict (keys
        timeScale_ms, timeData, f_MHz, Pxx, hist, hist_edges, segments) and kept in self.probe.  The three
        panels are drawn like the reference does when matplotlib is installed.  `fileNameStr` may also be a
        DeviceSignal (a window of a record already resident in HBM)."""
        from . import engine
        from .record import DeviceSignal
        if fileNameStr is None:
            fileNameStr = self.fileName
        samplesPerCode = self.samplesPerCode
        n_want = 10 * samplesPerCode
        ctx = engine.get_context(self, device)
        if isinstance(fileNameStr, DeviceSignal):
            sig = fileNameStr
            n = min(n_want, sig.length)
            head = sig.record.download(sig.offset + 1, max(0, min(samplesPerCode // 50, n) - 1))
            f, pxx, hist, nseg = ctx.probe_stats(sig.record, sig.offset, n, self.samplingFreq / 1000000.0)
        else:
            if not isinstance(fileNameStr, str):
                raise TypeError('File name must be a string')
            try:
                with open(fileNameStr, 'rb') as fid:
                    fid.seek(self.skipNumberOfBytes, 0)
                    data = np.fromfile(fid, self.dataType, n_want)
            except IOError as e:
                print('Unable to read file "%s": %s' % (fileNameStr, e))
                return None
            rec = ctx.upload(data)
            try:
                f, pxx, hist, nseg = ctx.probe_stats(rec, 0, data.size, self.samplingFreq / 1000000.0)
            finally:
                rec.free()
            head = data[1:samplesPerCode // 50]
        timeScale = np.arange(0, 0.005, 1 / self.samplingFreq)
        self.probe = dict(timeScale_ms=1000 * timeScale[1:samplesPerCode // 50], timeData=head,
                          f_MHz=f, Pxx=pxx, hist=hist, hist_edges=np.arange(-128, 128), segments=nseg)
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            return self.probe
        plt.figure(100)
        plt.clf()
        plt.subplot(2, 2, 1)
        if head is not None:
            plt.plot(self.probe["timeScale_ms"], head)
        plt.grid()
        plt.title('Time domain plot')
        plt.xlabel('Time (ms)')
        plt.ylabel('Amplitude')
        plt.subplot(2, 2, 2)
        plt.semilogy(f, pxx)
        plt.grid()
        plt.title('Frequency domain plot')
        plt.xlabel('Frequency (MHz)')
        plt.ylabel('Magnitude')
        plt.subplot(2, 1, 2)
        plt.bar(np.arange(-128, 127), hist, width=1.0, align='edge')
        plt.grid(True)
        plt.title('Histogram')
        plt.xlabel('Bin')
        plt.ylabel('Number in bin')
        return self.probe

    def postProcessing(self, fileNameStr=None):
        """acquire -> preRun -> track -> postNavigate on a record file: the call sequence of reference
        initialize.py:420-515 without the plots and without the .npy cache of the tracking results.
        Returns (acqResults, trackResults, navResults); navResults.solutions is unset when the record is too short
        or too few satellites carry ephemerides, as in the reference."""
        from . import acquisition, postNavigation, tracking
        print('Starting processing...')
        name = self.fileName if not fileNameStr else fileNameStr
        if not isinstance(name, str):
            raise TypeError('File name must be a string')
        if self.skipAcquisition:
            # (the reference then reads acqResults before anything assigned it: NameError, initialize.py:476,490)
            raise ValueError('skipAcquisition is set, but there are no acquisition results to reuse: '
                             'postProcessing() always acquires (initialize.py:476-490)')
        with open(name, 'rb') as fid:
            fid.seek(self.skipNumberOfBytes, 0)
            data = np.fromfile(fid, self.dataType, 11 * self.samplesPerCode)
            print('   Acquiring satellites...')
            acqResults = acquisition.AcquisitionResult(self)
            acqResults.acquire(data)
            if not np.any(acqResults.carrFreq):
                print('No GNSS signals detected, signal processing finished.')
                return acqResults, None, None
            acqResults.preRun()
            acqResults.showChannelStatus()
            trackResults = tracking.TrackingResult(acqResults)
            start = datetime.datetime.now()
            print('   Tracking started at %s' % start.strftime('%X'))
            trackResults.track(fid)
            self.lastTrackingSeconds = (datetime.datetime.now() - start).total_seconds()
            print('   Tracking is over (elapsed time %s s)' % self.lastTrackingSeconds)
        if not trackResults.has_results():   # (the reference's short-read exit: results were not set, tracking.py:159-163)
            return acqResults, trackResults, None
        print('   Calculating navigation solutions...')
        navResults = postNavigation.NavigationResult(trackResults)
        navResults.postNavigate()
        print('   Processing is complete for this data block')
        return acqResults, trackResults, navResults
