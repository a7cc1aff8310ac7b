"""TrackingResult with the reference's interface (reference tracking.py:6-295).

track(fid) reads the part of the record the channels need, puts it in HBM and runs every
channel's DLL/PLL loop in ONE launch of the persistent tracking kernel (sgx_track); the 13
per-millisecond series come back as a record array shaped like the reference's.
"""
from __future__ import print_function

import os

import numpy as np

from . import _native, engine
from .initialize import Result
from .record import DeviceFile

RESULT_DTYPE = [('status', 'S1'), ('absoluteSample', 'object'), ('codeFreq', 'object'),
                ('carrFreq', 'object'), ('I_P', 'object'), ('I_E', 'object'), ('I_L', 'object'),
                ('Q_E', 'object'), ('Q_P', 'object'), ('Q_L', 'object'), ('dllDiscr', 'object'),
                ('dllDiscrFilt', 'object'), ('pllDiscr', 'object'), ('pllDiscrFilt', 'object'),
                ('PRN', 'int64')]   # reference tracking.py:285-293


class TrackingResult(Result):
    def __init__(self, acqResult, device=None, verbose=False):
        """verbose: print the reference's progress lines (tracking.py:137-143: one per channel and 50 ms) - after the
        kernel has finished; the GPU is never synchronised every 50 ms to print."""
        self._verbose = verbose
        self._results = None
        self._acq = acqResult
        # (a deferred acquisition whose preRun is still to run on the device: the table arrives with the tracking results)
        self._channels = None if getattr(acqResult, "_prerun_pending", False) else acqResult.channels
        self._settings = acqResult.settings
        self._device = device
        self._lazy = None           # (series, active channel numbers): .results is packed on first access
        self.chained = False        # the last track() ran preRun on the device behind a deferred acquisition
        self.series = None          # float64[n_active, 13, ms] in _native.SERIES order
        self.kernel_ms = None       # HIP-event duration of the tracking kernel

    def has_results(self):
        """False after the reference's short-read exit (tracking.py:159-163 leaves the results unset) and before track()."""
        return self._results is not None or self._lazy is not None

    def _materialize(self):
        if self._channels is None and self._acq is not None:
            self._channels = self._acq.channels
        if self._lazy is not None:
            series, active = self._lazy
            self._lazy = None
            channel = self._channels
            # reference tracking.py:280-294: one record per ACTIVE channel, each series an object field
            res = np.recarray((len(active),), dtype=RESULT_DTYPE)
            for j, i in enumerate(active):
                res[j].status = channel[i].status
                res[j].PRN = int(channel[i].PRN)
                for k, name in enumerate(_native.SERIES):
                    res[j][name] = series[j, k]
            self._results = res

    def _data_type(self):
        """Settings.dataType -> (sgx_track_ex data_type, bytes per sample).  The reference reads np.fromfile(fid,
        dataType, blksize) (tracking.py:154) but seeks and tells in BYTES (tracking.py:107, 255): both are kept."""
        dt = np.dtype(self._settings.dataType)
        codes = {'i1': _native.DT_INT8, 'u1': _native.DT_UINT8, 'i2': _native.DT_INT16, 'u2': _native.DT_UINT16,
                 'i4': _native.DT_INT32, 'u4': _native.DT_UINT32, 'i8': _native.DT_INT64, 'u8': _native.DT_UINT64,
                 'f2': _native.DT_FLOAT16, 'f4': _native.DT_FLOAT32, 'f8': _native.DT_FLOAT64}
        key = dt.kind + str(dt.itemsize)
        # int8 / uint8 / int16 have their own kernels; float32 records of integers times one power of two are narrowed
        # to them exactly; every other real type is read sample by sample where it lies (include/sgx.h, sgx_track_ex)
        little = dt.byteorder in ('<', '|') or (dt.byteorder == '=' and np.little_endian)
        if key in codes and little:
            return codes[key], dt.itemsize
        raise TypeError("the GPU path tracks little-endian real IF samples - int8 ... uint64, float16 / 32 / 64 - not "
                        "Settings.dataType %r" % (self._settings.dataType,))

    def _window(self, fid, first, need):
        """Bytes [first, first+need) of the reference's file, as an HBM record."""
        ctx = engine.get_context(self._settings, self._device)
        dtype_code, isz = self._data_type()
        # a real file on disk: stream it natively (pinned double buffering, no numpy copy of the record), and let
        # tracking start while the transfer is still running
        name = getattr(fid, 'name', None)
        if isinstance(name, (str, bytes)) and os.path.isfile(name):
            return ctx.open_file(name, first, need)      # fills in the background; the kernel follows the watermark
        fid.seek(first, 0)
        if hasattr(fid, 'fileno'):
            try:
                data = np.fromfile(fid, np.int8, need)
            except (OSError, ValueError, AttributeError):
                data = np.frombuffer(fid.read(need), dtype=np.int8)
        else:
            data = np.frombuffer(fid.read(need), dtype=np.int8)
        return ctx.upload(data)

    def track(self, fid):
        """Code and carrier tracking of all channels (reference tracking.py:13-295).

        fid   open binary file of Settings.dataType samples (any real little-endian type; seek/read/tell/close), or a
              DeviceFile over a record already in HBM (for int16: the file's bytes, Context.upload_bytes).  Each active channel starts at byte
              skipNumberOfBytes + codePhase (tracking.py:107).
        On a short record the reference prints a message, closes fid and returns None without
        setting results (tracking.py:159-163); so does this method.
        """
        settings = self._settings
        ctx = engine.get_context(settings, self._device)
        ms = int(settings.msToProcess)            # float in the reference (Q9)
        nch = int(settings.numberOfChannels)
        self._lazy = None
        self._results = None
        self.series = None
        self.chained = False
        if self._channels is None and self._try_chained(fid, ctx, nch, ms):
            return
        if self._channels is None:
            self._channels = self._acq.channels   # (the queued sequence did not apply: the search is looked at, preRun runs here)
        channel = self._channels
        active = [i for i in range(nch) if channel[i].PRN != 0]
        if not active:
            self._results = np.recarray((0,), dtype=RESULT_DTYPE)
            self.series = np.empty((0, _native.NUM_SERIES, ms))
            return
        chans = [(int(channel[i].PRN), float(channel[i].acquiredFreq), float(channel[i].codePhase))
                 for i in active]
        dtype_code, isz = self._data_type()
        own = None
        if isinstance(fid, DeviceFile):
            rec, file_off = fid.record, fid.file_offset
        else:
            n = settings.samplesPerCode
            first = int(settings.skipNumberOfBytes + min(c[2] for c in chans))
            last = int(settings.skipNumberOfBytes + max(c[2] for c in chans))
            # the window starts on a 4 KiB boundary of the FILE: page-aligned reads, and every sample keeps the place
            # inside its 16-byte group that it has in a record resident from file offset 0 - the kernel's lanes then
            # add the same samples in the same order, and the series are bit-identical to the resident run's
            first_al = (first // 4096) * 4096
            need = (last - first_al) + (ms * (n + 2) + n) * isz   # a block is at most samplesPerCode + 1 long
            first = first_al
            own = rec = self._window(fid, first, need)
            file_off = first
        try:
            series, done = ctx.track(rec, chans, ms, rec_file_offset=file_off, data_type=dtype_code)
            self.kernel_ms = ctx.timing()["track_ms"]
        finally:
            if own is not None:
                own.free()
        if np.any(done != ms):
            print('Not able to read the specified number of samples for tracking, exiting!')
            fid.close()
            return None
        if self._verbose:
            for j, i in enumerate(active):
                for k in range(0, ms, 50):
                    print('Tracking: Ch %d' % (i + 1) + ' of %d' % nch + '; PRN#%02d' % int(channel[i].PRN) +
                          '; Completed %d' % k + ' of %d' % ms + ' msec')
        fid.seek(int(series[-1, 0, ms - 1]), 0)    # where the reference's last read left the file
        self.series = series
        self._lazy = (series, active)              # the record array of object fields is packed on first access of .results
        return

    def _try_chained(self, fid, ctx, nch, ms):
        """A deferred acquisition + preRun are pending on this context and the record is resident: preRun runs on the
        device, the tracking kernel behind it, and the host waits once (sgx_track_chained).  False: not applicable -
        the eager sequence follows."""
        acq = self._acq
        pend = getattr(acq, "_pending", None)
        if not (isinstance(fid, DeviceFile) and pend is not None and pend[0] is ctx and getattr(acq, "_prerun_pending", False)
                and getattr(ctx, "_acq_token", None) == pend[2] and not self._verbose and 1 <= nch <= 32):
            return False
        try:
            dtype_code, isz = self._data_type()
        except TypeError:
            return False
        got = ctx.track_chained(fid.record, nch, ms, rec_file_offset=fid.file_offset, data_type=dtype_code)
        if got is None:
            return False
        out, done, prn, freq, cph, n_act = got
        self.chained = True
        self.kernel_ms = ctx.timing()["track_ms"]
        acq._channels_from_table(prn, freq, cph, n_act)   # what preRun made, on the device
        self._channels = acq._channels
        if n_act == 0:
            self._results = np.recarray((0,), dtype=RESULT_DTYPE)
            self.series = np.empty((0, _native.NUM_SERIES, ms))
            return True
        if np.any(done[:n_act] != ms):
            print('Not able to read the specified number of samples for tracking, exiting!')
            fid.close()
            return True
        series = out[:n_act]
        fid.seek(int(series[-1, 0, ms - 1]), 0)
        self.series = series
        self._lazy = (series, list(range(n_act)))
        return True

    def plot(self):
        """One figure per tracked channel: discrete-time scatter, navigation bits, raw and filtered discriminators,
        correlator magnitudes (the panels of reference tracking.py:297-426).  Needs matplotlib; prints a notice
        and returns without it."""
        from .initialize import _pyplot
        plt = _pyplot("TrackingResult.plot")
        if plt is None:
            return
        assert isinstance(self._results, np.recarray)
        t = np.arange(int(self._settings.msToProcess)) / 1000.0
        for k, r in enumerate(self._results):
            plt.figure(200 + k)
            plt.clf()
            plt.suptitle('Channel %d (PRN %d) results' % (k, int(r.PRN)))
            ax = plt.subplot(3, 3, 1)
            ax.plot(r.I_P, r.Q_P, '.')
            ax.set_title('Discrete-Time Scatter Plot')
            ax.set_xlabel('I prompt')
            ax.set_ylabel('Q prompt')
            ax.axis('equal')
            ax = plt.subplot(3, 3, (2, 3))
            ax.plot(t, r.I_P)
            ax.set_title('Bits of the navigation message')
            ax.set_xlabel('Time (s)')
            for pos, series, title in ((4, r.pllDiscr, 'Raw PLL discriminator'), (7, r.pllDiscrFilt, 'Filtered PLL discriminator'),
                                       (6, r.dllDiscr, 'Raw DLL discriminator'), (9, r.dllDiscrFilt, 'Filtered DLL discriminator')):
                ax = plt.subplot(3, 3, pos)
                ax.plot(t, series)
                ax.set_title(title)
                ax.set_xlabel('Time (s)')
            ax = plt.subplot(3, 3, (5, 8))
            ax.plot(t, np.sqrt(r.I_E ** 2 + r.Q_E ** 2), t, np.sqrt(r.I_P ** 2 + r.Q_P ** 2),
                    t, np.sqrt(r.I_L ** 2 + r.Q_L ** 2), '-*')
            ax.set_title('Correlation results')
            ax.set_xlabel('Time (s)')
            ax.legend(['E', 'P', 'L'])
