"""Host logic of the QUEUED step (AcquisitionResult(deferred=True) -> preRun -> TrackingResult.track) without a GPU: the
state machine around sgx_acquire_begin / sgx_track_chained / sgx_acquire_end, against a stand-in context that answers with
the reference-made acquisition golden.  What the device does behind these calls is tests/test_gpu_parity.py's
(test_deferred_step_equals_the_eager_one, -m gpu)."""
import numpy as np
import pytest

from conftest import load_golden, pkg


class FakeCtx(object):
    """Answers the three queued-step calls of _native.Context; `chained` decides whether the device-side sequence applies."""

    def __init__(self, chained=True, n_detected=8):
        self.g = load_golden("acq_default.npz")
        self.chained = chained
        self.calls = []
        self._acq_token = 0
        self.n_detected = n_detected

    def acquire_begin(self, rec, offset, n, prn0, n_blocks=2, noncoh=False):
        self.calls.append("begin")
        self._prn = list(prn0)
        self._acq_token += 1
        return self._acq_token

    def acquire_end(self, n):
        self.calls.append("end")
        g = self.g
        idx = np.asarray(self._prn)
        return dict(carrFreq=g["carrFreq"][idx], codePhase=g["codePhase"][idx], peakMetric=g["peakMetric"][idx],
                    freqBin=g["freqBin"][idx], fineIdx=g["fineIdx"][idx])

    def acquire(self, rec, offset, n, prn0, n_blocks=2, noncoh=False):
        self.calls.append("acquire")
        self._acq_token += 1
        self._prn = list(prn0)
        return self.acquire_end(len(prn0))

    def track_chained(self, rec, n_ch, ms, rec_file_offset=0, data_type=0):
        self.calls.append("chained")
        if not self.chained:
            return None
        g = self.g
        out = np.zeros((n_ch, 13, ms))
        n_act = min(n_ch, self.n_detected)
        for i in range(n_act):
            out[i, 0] = g["ch_codePhase"][i] + 38192.0 * np.arange(1, ms + 1)
            out[i, 3] = 100.0 + i
        prn = np.zeros(n_ch, dtype=np.int32)
        freq = np.zeros(n_ch)
        cph = np.zeros(n_ch)
        prn[:n_act] = g["ch_PRN"][:n_act]
        freq[:n_act] = g["ch_acquiredFreq"][:n_act]
        cph[:n_act] = g["ch_codePhase"][:n_act]
        return out, np.full(n_ch, ms, dtype=np.int32), prn, freq, cph, n_act

    def track(self, rec, chans, ms, rec_file_offset=0, data_type=0):
        self.calls.append("track")
        out = np.zeros((len(chans), 13, ms))
        for i, c in enumerate(chans):
            out[i, 0] = c[2] + 38192.0 * np.arange(1, ms + 1)
            out[i, 3] = 100.0 + i
        return out, np.full(len(chans), ms, dtype=np.int32)

    def timing(self):
        return dict(track_ms=1.0, acquire_ms=1.0)


@pytest.fixture
def fake(monkeypatch):
    m = pkg()

    def install(**kw):
        ctx = FakeCtx(**kw)
        monkeypatch.setattr(m.engine, "get_context", lambda s, d=None: ctx)
        return m, ctx
    return install


def _objects(m, ctx, nch=8, ms=20):
    s = m.Settings()
    s.numberOfChannels = nch
    s.msToProcess = float(ms)
    rec = type("Rec", (), {"__len__": lambda self: 11 * 38192 + 100})()
    a = m.AcquisitionResult(s, device=0, deferred=True)
    a.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
    return s, rec, a


def test_a_deferred_search_is_looked_at_by_the_first_access(fake):
    m, ctx = fake()
    s, rec, a = _objects(m, ctx)
    assert ctx.calls == ["begin"] and a._pending is not None
    g = ctx.g
    assert np.array_equal(a.peakMetric, g["peakMetric"]) and ctx.calls == ["begin", "end"]
    assert np.array_equal(a.carrFreq, g["carrFreq"]) and np.array_equal(a.internals["fineIdx"], g["fineIdx"])
    assert ctx.calls == ["begin", "end"]                       # one look
    a.preRun()                                                 # the search has been looked at: the host's preRun, at once
    assert not a._prerun_pending
    assert np.array_equal(a.channels.PRN, g["ch_PRN"]) and np.array_equal(a.channels.acquiredFreq, g["ch_acquiredFreq"])


def test_prerun_of_a_queued_search_waits_for_whoever_looks(fake):
    m, ctx = fake()
    s, rec, a = _objects(m, ctx)
    a.preRun()
    assert a._prerun_pending and ctx.calls == ["begin"]
    g = ctx.g
    assert np.array_equal(a.channels.codePhase, g["ch_codePhase"])        # the look: acquire_end, then the host's preRun
    assert ctx.calls == ["begin", "end"] and not a._prerun_pending
    assert [x.decode() if isinstance(x, bytes) else x for x in a.channels.status] == ['T'] * 8


def test_the_chained_track_takes_preruns_table_from_the_device(fake):
    m, ctx = fake(n_detected=5)
    s, rec, a = _objects(m, ctx, nch=8)
    a.preRun()
    t = m.TrackingResult(a, device=0)
    fid = m.DeviceFile(rec)
    t.track(fid)
    assert t.chained and ctx.calls == ["begin", "chained"]    # nobody has looked at the search
    assert t.series.shape == (5, 13, 20) and t.has_results()
    assert np.array_equal(a.channels.PRN[:5], ctx.g["ch_PRN"][:5]) and not np.any(a.channels.PRN[5:])
    assert list(a.channels.status) == ['T'] * 5 + ['-'] * 3
    assert fid.tell() == int(t.series[-1, 0, -1])
    res = t.results                                             # packed now: one record per ACTIVE channel
    assert len(res) == 5 and res[2].PRN == ctx.g["ch_PRN"][2] and np.array_equal(res[2].I_P, t.series[2, 3])
    assert np.array_equal(a.results.peakMetric, ctx.g["peakMetric"]) and ctx.calls[-1] == "end"


def test_where_the_queued_sequence_does_not_apply_the_eager_calls_run(fake):
    m, ctx = fake(chained=False)
    s, rec, a = _objects(m, ctx)
    a.preRun()
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    assert not t.chained and ctx.calls == ["begin", "chained", "end", "track"]
    assert len(t.results) == 8 and np.array_equal(a.channels.PRN, ctx.g["ch_PRN"])
    # more than 32 channels, a verbose tracker, a plain file: never asked
    for kw in (dict(nch=40), dict(nch=8)):
        m2, ctx2 = fake()
        s2, rec2, a2 = _objects(m2, ctx2, **kw)
        a2.preRun()
        t2 = m2.TrackingResult(a2, device=0, verbose=(kw["nch"] == 8))
        t2.track(m2.DeviceFile(rec2))
        assert not t2.chained and "chained" not in ctx2.calls


def test_a_later_search_on_the_context_supersedes_a_pending_one(fake):
    m, ctx = fake()
    s, rec, a = _objects(m, ctx)
    b = m.AcquisitionResult(s, device=0)                       # an eager search on the same context
    b.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
    assert np.array_equal(b.peakMetric, ctx.g["peakMetric"])
    with pytest.raises(RuntimeError):
        a.results
    # ... and a tracker built on the superseded search does not chain on the newer one's page
    c = m.AcquisitionResult(s, device=0, deferred=True)
    c.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
    c.preRun()
    d = m.AcquisitionResult(s, device=0, deferred=True)
    d.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
    t = m.TrackingResult(c, device=0)
    with pytest.raises(RuntimeError):
        t.track(m.DeviceFile(rec))
