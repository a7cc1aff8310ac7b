"""GPU parity: the HIP path (through the C-ABI / drop-in classes) against the goldens captured
from the reference and against the CPU oracle.  Run with `pytest -m gpu` on an MI355X.

Bars: acquisition codePhase / frequencyBinIndex / fftMaxIndex and carrFreq bit-exact,
peakMetric within 1e-9 relative (ratio of fp64 FFT outputs); tracking absoluteSample bit-exact,
every correlator series within 1e-6 * max(1, RMS|P|) of the reference (BASELINE.json north_star).
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import load_golden, pkg, scene_from_json
from oracle import softgnss_oracle as orc

pytestmark = pytest.mark.gpu

TRK_TOL = 1e-6


def _ctx(settings=None):
    m = pkg()
    s = settings or m.Settings()
    return m, s, m.engine.get_context(s, 0)


def _trk_err(got, want):
    """max |delta| of the six correlator series over max(1, RMS(sqrt(I_P^2+Q_P^2))) per channel."""
    errs = []
    for c in range(want.shape[0]):
        scale = max(1.0, float(np.sqrt(np.mean(want[c, 3] ** 2 + want[c, 7] ** 2))))
        errs.append(np.max(np.abs(got[c, 3:9] - want[c, 3:9])) / scale)
    return max(errs)


def test_library_is_the_hip_build():
    m = pkg()
    assert os.path.exists(m._native.LIB_PATH)
    assert m._native.device_count() >= 1
    assert b"gfx950" in m._native.lib().sgx_version()


def test_device_generator_matches_host_generator():
    m, s, ctx = _ctx()
    sc = m.synth.Scene.default()
    for off, n in ((0, 100003), (123457, 65536), (38192 * 1000 + 5, 40000)):
        rec = ctx.synth(sc, n, offset=off)
        assert np.array_equal(rec.download(), m.synth.generate(sc, n, offset=off))
        rec.free()


def test_acquire_prn1_golden(default_record):
    g = load_golden("acq_prn1.npz")
    m = pkg()
    s = m.Settings()
    s.acqSatelliteList = [1]
    a = m.AcquisitionResult(s, device=0)
    a.acquire(default_record[:int(g["n_samples"])])
    assert np.array_equal(a.codePhase, g["codePhase"])
    assert np.array_equal(a.carrFreq, g["carrFreq"])
    assert a.internals["freqBin"][0] == g["freqBin"][0]
    assert a.internals["fineIdx"][0] == g["fineIdx"][0]
    assert np.allclose(a.peakMetric, g["peakMetric"], rtol=1e-9, atol=0)


def test_acquire_all_prns_and_prerun_golden(default_record):
    g = load_golden("acq_default.npz")
    m = pkg()
    s = m.Settings()
    a = m.AcquisitionResult(s, device=0)
    a.acquire(default_record[:int(g["n_samples"])])
    assert np.array_equal(a.codePhase, g["codePhase"])
    assert np.array_equal(a.carrFreq, g["carrFreq"])
    assert np.array_equal(a.internals["freqBin"], g["freqBin"])
    det = g["carrFreq"] > 0
    assert det.sum() == 8
    assert np.array_equal(a.internals["fineIdx"][det], g["fineIdx"][det])
    assert np.allclose(a.peakMetric, g["peakMetric"], rtol=1e-9, atol=0)
    a.preRun()
    assert np.array_equal(a.channels.PRN, g["ch_PRN"])
    assert np.array_equal(a.channels.acquiredFreq, g["ch_acquiredFreq"])
    assert np.array_equal(a.channels.codePhase, g["ch_codePhase"])
    assert [str(x) for x in a.channels.status] == [str(x) for x in g["ch_status"]]


def test_acquire_search_bands_and_prn_lists_against_oracle(default_record):
    """The default front end (the four-step kernels, their PRN chunks and tile order) with other Doppler grids and
    satellite lists than the default's 29 bins x 32 PRNs: 17, 41 and 57 bins (rows per PRN that do and do not divide
    the chunk, row counts that are and are not multiples of the XCD count), a list of 5 PRNs, and the 10-ms non-coherent
    extension on a narrow band - codePhase, carrFreq and the bins exactly, peakMetric to 1e-9."""
    m = pkg()
    n = 38192
    for band, prns, nb, nc in ((8.0, None, 2, False), (20.0, None, 2, False), (28.0, [1, 3, 7, 11, 30], 2, False),
                               (6.0, [3, 7, 14, 19, 22, 25, 31], 10, True)):
        s = m.Settings()
        so = orc.OracleSettings()
        for o in (s, so):
            o.acqSearchBand = band
            if prns is not None:
                o.acqSatelliteList = prns
        x = default_record[:(10 + nb) * n if nc else 11 * n]
        a = m.AcquisitionResult(s, device=0)
        a.acquire(x, n_blocks=nb, noncoh=nc)
        w = orc.acquire(so, x, n_blocks=nb, noncoh=nc) if nc else orc.acquire(so, x)
        assert np.array_equal(a.codePhase, w["codePhase"]), band
        assert np.array_equal(a.carrFreq, w["carrFreq"]), band
        assert np.allclose(a.peakMetric, w["peakMetric"], rtol=1e-9, atol=0), band


def test_acquire_code_phase_edges_golden():
    g = load_golden("acq_edges.npz")
    m = pkg()
    s = m.Settings()
    s.acqSatelliteList = [1]
    for i, c in enumerate(g["phases"]):
        x = m.synth.generate(scene_from_json(g["scenes"][i]), 11 * s.samplesPerCode)
        a = m.AcquisitionResult(s, device=0)
        if str(g["err"][i]) == "IndexError":
            with pytest.raises(IndexError):
                a.acquire(x)
            continue
        a.acquire(x)
        assert a.codePhase[0] == g["codePhase"][i] == c
        assert a.carrFreq[0] == g["carrFreq"][i]
        assert a.internals["freqBin"][0] == g["freqBin"][i]
        assert a.internals["fineIdx"][0] == g["fineIdx"][i]
        assert np.isclose(a.peakMetric[0], g["peakMetric"][i], rtol=1e-9, atol=0)


def test_acquire_noncoherent_extension_vs_oracle(default_record):
    """BASELINE.json config 4 shape (non-coherent sum over blocks), small: 3 PRNs x 4 blocks."""
    m = pkg()
    s = m.Settings()
    s.acqSatelliteList = [1, 2, 3]
    a = m.AcquisitionResult(s, device=0)
    a.acquire(default_record[:14 * 38192], n_blocks=4, noncoh=True)
    os_ = orc.OracleSettings(acqSatelliteList=[1, 2, 3])
    r = orc.acquire(os_, default_record[:14 * 38192], n_blocks=4, noncoh=True)
    assert np.array_equal(a.codePhase, r["codePhase"])
    assert np.array_equal(a.carrFreq, r["carrFreq"])
    assert np.array_equal(a.internals["freqBin"][:3], r["freqBin"][:3])
    assert np.allclose(a.peakMetric, r["peakMetric"], rtol=1e-9, atol=0)


def test_acquire_device_signal_equals_host_signal(default_record):
    m, s, ctx = _ctx()
    g = load_golden("acq_prn1.npz")
    sc = scene_from_json(g["scene"])
    rec = ctx.synth(sc, 11 * 38192)
    s1 = m.Settings()
    s1.acqSatelliteList = [1]
    a = m.AcquisitionResult(s1, device=0)
    a.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
    assert np.array_equal(a.codePhase, g["codePhase"]) and np.array_equal(a.carrFreq, g["carrFreq"])


def test_acquire_variants_agree(default_record):
    """The knobs that pick another implementation of a stage: the coarse outcome fetched by a stream synchronisation
    instead of the spin on the pinned page (SGX_ACQ_SPIN=0), the fine search on the pass-per-radix transform
    (SGX_ACQ_FINE_V1=1), the round-1 correlation that mixes every Doppler bin (SGX_ACQ_V1=1), the host deciding the
    detections between the coarse and the fine search (SGX_ACQ_DEVICE_LED=0), the call's front as four launches
    (SGX_ACQ_FRONT=0), the round-4 second-peak search that transforms the winning row again instead of reading the
    per-residue (maximum, maximum of the others) pairs of the one pass (SGX_ACQ_TOP2=0), the correlation batch on one queue
    instead of two (SGX_ACQ_STREAMS=1): same indices, same frequencies, peak metrics equal to rounding (the last four: bit
    for bit)."""
    g = load_golden("acq_default.npz")
    m = pkg()
    s = m.Settings()
    x = default_record[:int(g["n_samples"])]
    ref = m.AcquisitionResult(s, device=0)
    ref.acquire(x)
    assert np.array_equal(ref.codePhase, g["codePhase"]) and np.array_equal(ref.carrFreq, g["carrFreq"])
    for env in ({"SGX_ACQ_SPIN": "0"}, {"SGX_ACQ_FINE_V1": "1"}, {"SGX_ACQ_V1": "1"}, {"SGX_ACQ_DEVICE_LED": "0"},
                {"SGX_ACQ_FRONT": "0"}, {"SGX_ACQ_DEVICE_LED": "0", "SGX_ACQ_SPIN": "0"}, {"SGX_ACQ_TOP2": "0"},
                {"SGX_ACQ_TOP2": "0", "SGX_ACQ_DEVICE_LED": "0"}, {"SGX_ACQ_STREAMS": "1"}):
        os.environ.update(env)
        try:
            a = m.AcquisitionResult(s, device=0)
            a.acquire(x)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert np.array_equal(a.codePhase, ref.codePhase) and np.array_equal(a.carrFreq, ref.carrFreq), env
        assert np.array_equal(a.internals["freqBin"], ref.internals["freqBin"]), env
        assert np.allclose(a.peakMetric, ref.peakMetric, rtol=1e-9, atol=0), env
        if "SGX_ACQ_DEVICE_LED" in env or "SGX_ACQ_FRONT" in env or "SGX_ACQ_TOP2" in env or "SGX_ACQ_STREAMS" in env:
            assert np.array_equal(a.peakMetric, ref.peakMetric), env


def _golden_tracker(m, g, ms=None, nch=4):
    s = m.Settings()
    s.numberOfChannels = nch
    s.msToProcess = float(int(g["ms"]) if ms is None else ms)
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([g["ch_PRN"][:nch], g["ch_acquiredFreq"][:nch], g["ch_codePhase"][:nch],
                                     ['T'] * nch], names='PRN,acquiredFreq,codePhase,status')
    return s, m.TrackingResult(a, device=0)


def test_track_golden_via_real_file(default_record):
    g = load_golden("trk_default.npz")
    m = pkg()
    s, t = _golden_tracker(m, g)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        default_record.tofile(f.name)
        with open(f.name, "rb") as fid:
            t.track(fid)
            assert fid.tell() == int(g["end_pos"])
    want = g["series"]
    got = t.series
    assert got.shape == want.shape
    assert np.array_equal(got[:, 0], want[:, 0])                     # absoluteSample bit-exact
    assert _trk_err(got, want) < TRK_TOL
    assert np.max(np.abs(got[:, 1] - want[:, 1])) < 1e-6             # codeFreq, Hz
    assert np.max(np.abs(got[:, 2] - want[:, 2])) < 1e-5             # carrFreq, Hz
    assert np.max(np.abs(got[:, 9:13] - want[:, 9:13])) < 1e-7       # discriminators / NCO commands
    r = t.results
    assert len(r) == 4 and list(r.PRN) == list(g["PRN"])
    assert r[0].status == b'T' and r.dtype.names[0] == 'status'
    assert np.array_equal(r[2].I_P, got[2, 3])


def test_track_progress_lines_behind_verbose(default_record, capsys):
    """The reference prints a progress line per channel and 50 ms (tracking.py:137-143); offered behind verbose=True,
    printed after the kernel has finished."""
    g = load_golden("trk_default.npz")
    m = pkg()
    s, t0 = _golden_tracker(m, g)
    s.msToProcess = 120.0
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload(default_record)
    t = m.TrackingResult(t0, device=0, verbose=True)
    capsys.readouterr()
    t.track(m.DeviceFile(rec))
    out = capsys.readouterr().out
    n_act = int(np.sum(t0.channels.PRN != 0))
    assert out.count("Tracking: Ch ") == 3 * n_act
    assert "Tracking: Ch 1 of %d; PRN#%02d; Completed 100 of 120 msec" % (s.numberOfChannels, int(t0.channels.PRN[0])) in out
    t2 = m.TrackingResult(t0, device=0)
    t2.track(m.DeviceFile(rec))
    assert "Tracking: Ch" not in capsys.readouterr().out


def test_track_device_file_equals_host_file(default_record):
    g = load_golden("trk_default.npz")
    m, s0, ctx = _ctx()
    s, t = _golden_tracker(m, g, ms=120)
    rec = ctx.upload(default_record)
    fid = m.DeviceFile(rec)
    t.track(fid)
    assert np.array_equal(t.series[:, 0], g["series"][:, 0, :120])
    assert _trk_err(t.series, g["series"][:, :, :120]) < TRK_TOL
    assert fid.tell() == int(g["series"][3, 0, 119])


def test_track_short_record_returns_none(default_record):
    g = load_golden("trk_default.npz")
    gs = load_golden("trk_short.npz")
    m = pkg()
    s, t = _golden_tracker(m, g)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        default_record[:int(gs["n_samples"])].tofile(f.name)
        fid = open(f.name, "rb")
        ret = t.track(fid)
        assert ret is None and t._results is None and fid.closed
        with pytest.raises(AssertionError):
            t.results


def test_track_inactive_channels_and_skip_bytes(default_record):
    """Channels with PRN 0 produce no record (Q8); skipNumberOfBytes shifts file positions."""
    g = load_golden("trk_default.npz")
    m = pkg()
    s = m.Settings()
    s.numberOfChannels = 6
    s.msToProcess = 50.0
    s.skipNumberOfBytes = 1000
    a = m.AcquisitionResult(s, device=0)
    prn = np.r_[g["ch_PRN"][:2], 0, 0, 0, 0]
    a._channels = np.rec.fromarrays([prn, np.r_[g["ch_acquiredFreq"][:2], 0, 0, 0, 0],
                                     np.r_[g["ch_codePhase"][:2], 0, 0, 0, 0], ['T', 'T', '-', '-', '-', '-']],
                                    names='PRN,acquiredFreq,codePhase,status')
    t = m.TrackingResult(a, device=0)
    padded = np.r_[np.zeros(1000, dtype=np.int8), default_record[:60 * 38192]]
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        padded.tofile(f.name)
        with open(f.name, "rb") as fid:
            t.track(fid)
    assert len(t.results) == 2
    want = g["series"][:2, :, :50]
    assert np.array_equal(t.series[:, 0], want[:, 0] + 1000)
    assert _trk_err(t.series, want) < TRK_TOL


def test_track_replicated_channels_identical(default_record):
    """BASELINE.json config 5 shape: replicas of one channel init give bit-identical series."""
    g = load_golden("trk_default.npz")
    m, s0, ctx = _ctx()
    rec = ctx.upload(default_record[:70 * 38192])
    chans = [(int(g["ch_PRN"][i % 4]), float(g["ch_acquiredFreq"][i % 4]), float(g["ch_codePhase"][i % 4]))
             for i in range(16)]
    series, done = ctx.track(rec, chans, 60)
    assert np.all(done == 60)
    for i in range(4, 16):
        assert np.array_equal(series[i], series[i % 4])
    assert _trk_err(series[:4], g["series"][:, :, :60]) < TRK_TOL


# ---- full-size, size-independent properties (BASELINE.json config 3 / config 5 shapes) ---------------

@pytest.fixture(scope="module")
def full_run():
    """8 channels x 37 000 ms on a 1.41 GB record generated in HBM; default cooperative split."""
    m = pkg()
    s = m.Settings()
    ctx = m.engine.get_context(s, 0)
    sc = m.synth.Scene.default()
    rec = ctx.synth(sc, m.synth.record_length(s.samplesPerCode, 37000))
    a = m.AcquisitionResult(s, device=0)
    a.acquire(m.DeviceSignal(rec, 0, 11 * s.samplesPerCode))
    a.preRun()
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    series, done = ctx.track(rec, chans, 37000)
    yield m, s, ctx, sc, rec, a, chans, series, done
    rec.free()


def test_full_length_tracking_locks_onto_the_scene(full_run):
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    assert np.all(done == 37000)
    truth = {sat["prn"]: sat for sat in sc.sats}
    for i, (prn, f0, cp) in enumerate(chans):
        pos = series[i, 0]
        d = np.diff(np.r_[cp, pos])
        assert set(np.unique(d)).issubset({38191.0, 38192.0, 38193.0})       # block sizes
        sat = truth[prn]
        f_true = sat["car_fcw"] / 2.0 ** 32 * s.samplingFreq
        assert abs(np.mean(series[i, 2, 200:]) - f_true) < 1.0                # carrier NCO on the truth (Hz)
        c_true = sat["code_fcw"] / 2.0 ** 32 * s.samplingFreq
        assert abs(np.mean(series[i, 1, 2000:]) - c_true) < 0.2               # code NCO on the truth (Hz)
        ip, qp = series[i, 3, 500:], series[i, 7, 500:]
        assert np.mean(np.abs(ip)) > 8 * np.mean(np.abs(qp))                  # phase lock: energy in I_P
        # nav-bit sign changes of I_P happen only on a 20 ms raster
        flips = np.flatnonzero(np.diff(np.sign(ip)) != 0)
        assert len(flips) > 100 and len(np.unique(flips % 20)) == 1
    # total streamed bytes = SURVEY section 8(d) algorithmic figure for config 3
    streamed = sum(series[i, 0, -1] - chans[i][2] for i in range(8))
    assert abs(streamed / (8 * 37000 * 38192.0) - 1.0) < 1e-4


def test_split_variants_agree_on_the_full_run(full_run):
    """The decompositions of the round-3 latency-mode kernel against the default (the speculative kernel, 20 members per
    channel): 30 members per channel (one per unit and correlator arm), one workgroup per channel owning all ten units,
    ten members of one unit with all three arms and the placement-independent exchange path, three members owning
    four / three / three units, and the arm split switched off.  Identical block boundaries, sums equal to rounding
    (different summation order only)."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    assert ctx.timing()["track_kernel"] == 5 and ctx.timing()["track_members"] == 20      # (what full_run ran)
    old = {k: os.environ.get(k) for k in ("SGX_TRK_SPLIT", "SGX_TRK_FASTX", "SGX_TRK_ARMS", "SGX_TRK_V3")}
    try:
        for env, members in (({"SGX_TRK_V3": "0"}, 30), ({"SGX_TRK_SPLIT": "1"}, 1), ({"SGX_TRK_SPLIT": "10", "SGX_TRK_FASTX": "0"}, 10),
                             ({"SGX_TRK_SPLIT": "3"}, 3), ({"SGX_TRK_ARMS": "3"}, 10)):
            os.environ.update(env)
            ms = 6000
            s2, d2 = ctx.track(rec, chans, ms)
            tm = ctx.timing()
            assert tm["track_kernel"] == 2 and tm["track_members"] == members, (env, tm)
            assert np.all(d2 == ms)
            assert np.array_equal(s2[:, 0], series[:, 0, :ms])
            assert _trk_err(s2, series[:, :, :ms]) < 1e-9
            for k in env:
                os.environ.pop(k)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_cooperative_timeout_falls_back_to_one_workgroup_per_channel(full_run, capfd):
    """A member that times out flags the channel; the host repeats the launch with split = 1 (same results)."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    os.environ["SGX_TRK_TEST_TIMEOUT"] = "1"
    try:
        s2, d2 = ctx.track(rec, chans, 500)
    finally:
        os.environ.pop("SGX_TRK_TEST_TIMEOUT", None)
    assert "repeating the launch with one workgroup per channel" in capfd.readouterr().err
    assert np.all(d2 == 500) and np.array_equal(s2[:, 0], series[:, 0, :500])
    assert _trk_err(s2, series[:, :, :500]) < 1e-9


def test_timeout_after_a_stalled_stream_still_falls_back(full_run, capfd):
    """Three launches at most: the first treated as a stalled stream (repeated with the same decomposition), the
    repeat treated as a member's timeout - the third must really run with one workgroup per channel and succeed
    (round 3 returned 'timeout with split 1' here without launching it)."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    os.environ["SGX_TRK_TEST_STALL"] = "1"
    os.environ["SGX_TRK_TEST_TIMEOUT"] = "2"
    try:
        s2, d2 = ctx.track(rec, chans, 500)
        members = ctx.timing()["track_members"]
    finally:
        os.environ.pop("SGX_TRK_TEST_STALL", None)
        os.environ.pop("SGX_TRK_TEST_TIMEOUT", None)
    err = capfd.readouterr().err
    assert "repeating the launch on the resident record" in err
    assert "repeating the launch with one workgroup per channel" in err
    assert members == 1
    assert np.all(d2 == 500) and np.array_equal(s2[:, 0], series[:, 0, :500])
    assert _trk_err(s2, series[:, :, :500]) < 1e-9


def test_many_channels_throughput_mode(full_run):
    """256 channels (32 replicas of the 8 inits) on one GPU: one CU per channel, replicas bit-identical."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    many = [chans[i % 8] for i in range(256)]
    ms = 300
    s2, d2 = ctx.track(rec, many, ms)
    assert np.all(d2 == ms)
    for i in range(8, 256):
        assert np.array_equal(s2[i], s2[i % 8])
    assert np.array_equal(s2[:8, 0], series[:, 0, :ms])
    assert _trk_err(s2[:8], series[:, :, :ms]) < 1e-9


def test_many_channels_at_staggered_offsets_against_the_oracle(full_run):
    """The shape the many-channel bench leg runs: throughput-mode channels (one workgroup per channel, trk_kernel_tp)
    that start at DIFFERENT places of the record.  256 channels = 32 groups of the 8 acquired inits, group j started j
    code periods (j x 38 192 samples) behind the acquired code phase - the code drifts by at most 0.1 chip in 31 ms, so
    every replica pulls in - and eight of them, one per PRN and each at another offset, are compared with the oracle
    itself: block boundaries identical, sums to 1e-9."""
    from concurrent.futures import ProcessPoolExecutor
    from oracle_helpers import oracle_channel
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    n = s.samplesPerCode
    ms = 300
    many = [(chans[i % 8][0], chans[i % 8][1], chans[i % 8][2] + (i // 8) * n) for i in range(256)]
    s2, d2 = ctx.track(rec, many, ms)
    tm = ctx.timing()
    assert tm["track_kernel"] == 3 and tm["track_members"] == 1 and np.all(d2 == ms)
    picks = [8 * (4 * j + 1) + j for j in range(8)]          # PRN j of group 4 j + 1: offsets 1, 5, .. 29 code periods
    assert len({many[i][2] - chans[i % 8][2] for i in picks}) == 8 and sorted(i % 8 for i in picks) == list(range(8))
    host = rec.download(0, (32 + ms + 3) * n)
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        want = np.stack(list(ex.map(oracle_channel, [(host,) + many[i] + (ms,) for i in picks])))
    got = s2[picks]
    assert np.array_equal(got[:, 0], want[:, 0])
    assert _trk_err(got, want) < 1e-9
    assert np.max(np.abs(got[:, 1] - want[:, 1])) < 1e-6 and np.max(np.abs(got[:, 2] - want[:, 2])) < 1e-5
    # a start shifted by whole code periods sees the same signal a few periods later: locked like the unshifted replica
    for i in picks:
        assert np.mean(np.abs(got[picks.index(i), 3, 100:])) > 8 * np.mean(np.abs(got[picks.index(i), 7, 100:]))


def test_one_workgroup_per_channel_agrees_on_a_long_run(full_run):
    """One workgroup per channel (every member-to-member exchange degenerate, all units in one map) against the default
    30-member launch over a long run including the pull-in transient: identical block boundaries, sums equal to
    rounding, NCO frequencies within 1e-6 Hz."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    old = os.environ.get("SGX_TRK_SPLIT")
    try:
        os.environ["SGX_TRK_SPLIT"] = "1"
        ms = 12000
        s2, d2 = ctx.track(rec, chans, ms)
    finally:
        if old is None:
            os.environ.pop("SGX_TRK_SPLIT", None)
        else:
            os.environ["SGX_TRK_SPLIT"] = old
    assert np.all(d2 == ms)
    assert np.array_equal(s2[:, 0], series[:, 0, :ms])
    assert _trk_err(s2, series[:, :, :ms]) < 1e-9
    assert np.max(np.abs(s2[:, 1:3] - series[:, 1:3, :ms])) < 1e-6


@pytest.mark.parametrize("layout", ["spec", "arms", "units"])
def test_a_withheld_member_aborts_the_channel_quickly(full_run, capfd, layout):
    """The launch really lacks one member per channel (SGX_TRK_TEST_WITHHOLD=1): the others must give the channel up
    within one poll budget - not spin block after block - and the host's repeat with one workgroup per channel must
    deliver the usual results.  The speculative kernel (20 members per channel) and both member layouts of the round-3
    latency-mode kernel (30 per channel: unit x arm; 10: units)."""
    import time
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    os.environ["SGX_TRK_TEST_WITHHOLD"] = "1"
    if layout == "units":
        os.environ["SGX_TRK_ARMS"] = "3"
    if layout == "arms":
        os.environ["SGX_TRK_V3"] = "0"
    try:
        t0 = time.perf_counter()
        s2, d2 = ctx.track(rec, chans, 500)
        dt = time.perf_counter() - t0
    finally:
        os.environ.pop("SGX_TRK_TEST_WITHHOLD", None)
        os.environ.pop("SGX_TRK_ARMS", None)
        os.environ.pop("SGX_TRK_V3", None)
    assert "repeating the launch with one workgroup per channel" in capfd.readouterr().err
    assert dt < 20.0, dt
    assert np.all(d2 == 500) and np.array_equal(s2[:, 0], series[:, 0, :500])
    assert _trk_err(s2, series[:, :, :500]) < 1e-9


@pytest.mark.parametrize("layout", ["spec", "arms", "units", "one"])
def test_a_block_beyond_the_units_of_the_launch_is_an_error_not_a_silent_truncation(layout):
    """A DLL bandwidth of 6 kHz on a channel without a signal drives the code NCO tens of kHz off: blocks become longer than
    the ten 4096-sample units of the launch.  Every member layout must say so (SGX_E_RANGE) instead of dropping the tail
    samples; at 2 kHz the same run still fits and completes."""
    m = pkg()
    chans = [(7, 9.548e6, 1234.0), (1, 9.5478e6, 12345.0)]
    if layout == "units":
        os.environ["SGX_TRK_ARMS"] = "3"
    if layout == "one":
        os.environ["SGX_TRK_SPLIT"] = "1"
    if layout == "arms":
        os.environ["SGX_TRK_V3"] = "0"
    try:
        for bw, fits in ((2000.0, True), (6000.0, False)):
            s = m.Settings()
            s.dllNoiseBandwidth = bw
            s.msToProcess = 60.0
            s.numberOfChannels = 2
            ctx = m.engine.get_context(s, 0)
            rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(s.samplesPerCode, 80))
            if fits:
                ser, done = ctx.track(rec, chans, 60)
                assert np.all(done == 60)
            else:
                with pytest.raises(m._native.SgxError, match="plausible range"):
                    ctx.track(rec, chans, 60)
    finally:
        os.environ.pop("SGX_TRK_ARMS", None)
        os.environ.pop("SGX_TRK_SPLIT", None)
        os.environ.pop("SGX_TRK_V3", None)


def test_second_front_end_golden():
    """Another samplesPerCode (16368 = 2^4*3*11*31: other FFT radices, 5 tracking units instead of 10)."""
    g = load_golden("rate2.npz")
    m = pkg()
    s = m.Settings()
    s.samplingFreq = 16367600.0
    s.IF = 4130400.0
    s.msToProcess = 250.0
    s.numberOfChannels = 3
    s.acqSatelliteList = range(1, 13)
    n = s.samplesPerCode
    assert n == 16368
    rec = m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]))
    a = m.AcquisitionResult(s, device=0)
    a.acquire(rec[:11 * n])
    assert np.array_equal(a.codePhase, g["codePhase"])
    assert np.array_equal(a.carrFreq, g["carrFreq"])
    assert np.array_equal(a.internals["freqBin"][:12], g["freqBin"][:12])
    assert np.allclose(a.peakMetric, g["peakMetric"], rtol=1e-9, atol=0)
    a.preRun()
    assert np.array_equal(a.channels.PRN, g["ch_PRN"])
    for env in ({}, {"SGX_TRK_SPLIT": "1"}, {"SGX_TRK_ARMS": "3"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            t = m.TrackingResult(a, device=0)
            with tempfile.NamedTemporaryFile(suffix=".bin") as f:
                rec.tofile(f.name)
                with open(f.name, "rb") as fid:
                    t.track(fid)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert np.array_equal(t.series[:, 0], g["series"][:, 0])
        assert _trk_err(t.series, g["series"]) < TRK_TOL


def test_streaming_file_ingest_matches_file_bytes():
    """sgx_if_upload_file: several 32 MiB chunks, an unaligned offset, a short file."""
    m, s, ctx = _ctx()
    rng = np.random.default_rng(5)
    data = rng.integers(-128, 128, size=(32 << 20) * 2 + 12345, dtype=np.int8)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        data.tofile(f.name)
        for off, n in ((0, data.size), (777, (32 << 20) + 99), (data.size - 1000, 5000)):
            rec = ctx.upload_file(f.name, off, n)
            want = data[off:off + n]
            assert len(rec) == want.size
            assert np.array_equal(rec.download(), want)
            rec.free()
    with pytest.raises(m._native.SgxError):
        ctx.upload_file("/nonexistent/record.bin", 0, 10)


# ---- next row: bit sync + preamble search on the tracking output -------------------

def _nav_settings(m, nch=2):
    s = m.Settings()
    s.samplingFreq = 16367600.0
    s.IF = 4130400.0
    s.msToProcess = 10000.0
    s.numberOfChannels = nch
    return s


def _nav_scene(m, g):
    return scene_from_json(g["scene"]).with_subframes(int(g["subframes_at"]))


def test_device_generator_matches_host_generator_with_subframes():
    g = load_golden("nav_preambles.npz")
    m, s, ctx = _ctx()
    sc = _nav_scene(m, g)
    for off, n in ((0, 70001), (16368 * 20 * 2047 + 11, 16368 * 45), (16368 * 1999 - 300, 50000)):
        rec = ctx.synth(sc, n, offset=off)
        assert np.array_equal(rec.download(), m.synth.generate(sc, n, offset=off))
        rec.free()


def test_find_preambles_on_reference_series():
    """Product vs the reference's own findPreambles result on the reference tracker's I_P (committed fixture),
    plus oracle agreement on perturbed copies (sign-scrambled channel, heavy noise, shifted search start)."""
    g = load_golden("nav_preambles.npz")
    m, s0, ctx = _ctx()
    got = ctx.find_preambles(g["I_P"])
    assert np.array_equal(got, g["firstSubFrame"])
    rng = np.random.default_rng(11)
    scr = g["I_P"][1] * rng.choice([-1.0, 1.0], size=g["I_P"].shape[1])
    noisy = g["I_P"][0] + rng.normal(0.0, 1.5 * np.std(g["I_P"][0]), size=g["I_P"].shape[1])
    x = np.stack([g["I_P"][0], scr, noisy, -g["I_P"][1], g["I_P"][0][:]])
    want, _ = orc.find_preambles(x, ['T'] * 5, 5)
    assert np.array_equal(ctx.find_preambles(x), want)
    assert want[0] == 1999 and want[1] == 0 and want[3] == 1999
    for start in (0, 1999, 2000, 2400):
        w2, _ = orc.find_preambles(x[:1], ['T'], 1, search_start=start)
        assert np.array_equal(ctx.find_preambles(x[:1], start), w2)
    # records cut short around a candidate: whatever the reference's numpy code does (result or exception type)
    full = g["I_P"][:1]
    for lo, hi in ((1979, 10000), (1960, 10000), (1959, 10000), (0, 9150), (2000, 9250), (2000, 9199),
                   (2000, 9200), (2000, 9210), (7000, 10000), (1990, 8630), (1990, 8640)):
        cut = full[:, lo:hi]
        try:
            want = orc.find_preambles(cut, ['T'], 1)[0]
        except (ValueError, IndexError) as e:
            with pytest.raises(type(e)):
                ctx.find_preambles(cut)
        else:
            assert np.array_equal(ctx.find_preambles(cut), want), (lo, hi)


def test_navigation_result_end_to_end_from_gpu_tracking():
    """acquire -> preRun -> track -> findPreambles, all through the product, on the structured-navigation scene
    the fixture was made from: same subframe start as the reference found on its own tracker's output."""
    g = load_golden("nav_preambles.npz")
    m = pkg()
    s = _nav_settings(m)
    ctx = m.engine.get_context(s, 0)
    sc = _nav_scene(m, g)
    rec = ctx.synth(sc, int(g["n_samples"]))
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([g["ch_PRN"], g["ch_acquiredFreq"], g["ch_codePhase"], ['T', 'T']],
                                    names='PRN,acquiredFreq,codePhase,status')
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    rec.free()
    ip = np.stack([np.asarray(t.results[k].I_P, dtype=np.float64) for k in range(2)])
    scale = np.sqrt(np.mean(g["I_P"] ** 2))
    assert np.max(np.abs(ip - g["I_P"])) / scale < TRK_TOL
    nav = m.NavigationResult(t, device=0)
    first, active = nav.findPreambles()
    assert np.array_equal(first, g["firstSubFrame"]) and np.array_equal(active, g["activeChnList"])
    # an inactive channel is skipped, and the reference's row-index quirk is kept (row 0 serves the only active one)
    t.results[0].status = b'-'
    first2, active2 = nav.findPreambles()
    w2, a2 = orc.find_preambles(ip, ['-', 'T'], 2)
    assert np.array_equal(first2, w2) and np.array_equal(active2, a2)


def test_nav_bits_recover_the_transmitted_subframes():
    """32.1 s of the structured scene: track, find the subframe start, integrate the bits (sgx_nav_bits) and
    compare with the oracle on the same I_P and with the bit table the generator transmitted."""
    g = load_golden("nav_preambles.npz")
    m = pkg()
    s = _nav_settings(m)
    s.msToProcess = 32100.0
    ctx = m.engine.get_context(s, 0)
    sc = _nav_scene(m, g)
    rec = ctx.synth(sc, m.synth.record_length(s.samplesPerCode, 32100))
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([g["ch_PRN"], g["ch_acquiredFreq"], g["ch_codePhase"], ['T', 'T']],
                                    names='PRN,acquiredFreq,codePhase,status')
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    rec.free()
    nav = m.NavigationResult(t, device=0)
    first, active = nav.findPreambles()
    assert list(first) == [1999, 1999] and list(active) == [0, 1]
    bits = nav.navBits(first, active)
    at = int(g["subframes_at"])
    for ch in (0, 1):
        ip = np.asarray(t.results[ch].I_P, dtype=np.float64)
        want = orc.nav_bits(ip, int(first[ch]))
        got = np.array([int(b) for b in bits[ch]])
        assert len(got) == 1501 and np.array_equal(got, want)
        sent = sc.nav_bits[ch][(at - 1 + np.arange(1501)) % m.synth.NAV_TABLE_BITS]
        assert np.array_equal(got, sent) or np.array_equal(got, 1 - sent)   # Costas loop: sign ambiguity


# ---- next row: probeData statistics ------------------------------------------------

PSD_TOL = 1e-9     # relative, per bin (fp64 FFT with another operation order than pocketfft)


def test_probe_statistics_golden():
    g = load_golden("probe_default.npz")
    m = pkg()
    s = m.Settings()
    data = m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]) + 999)
    with tempfile.NamedTemporaryFile(suffix=".bin") as fh:
        data.tofile(fh.name)
        out = s.probeData(fh.name, device=0)
    assert out is s.probe and out["segments"] == 24
    assert np.array_equal(out["hist"], g["hist"])
    assert np.array_equal(out["f_MHz"], g["f"])
    assert np.max(np.abs(out["Pxx"] - g["Pxx"]) / g["Pxx"]) < PSD_TOL
    assert np.array_equal(out["timeData"], g["time_amp"]) and np.allclose(out["timeScale_ms"], g["time_ms"], rtol=0, atol=0)
    # resident record, window at an offset, extreme sample values, another sampling rate
    ctx = m.engine.get_context(s, 0)
    rng = np.random.default_rng(21)
    x = rng.integers(-128, 128, size=500000, dtype=np.int8)
    x[1000:1200] = -128
    x[5000:5100] = 127
    rec = ctx.upload(x)
    for off, n in ((0, 381920), (12345, 300001), (7, 16384), (99, 16385 + 15359)):
        f, pxx, hist, nseg = ctx.probe_stats(rec, off, n, 16.3676)
        s2 = orc.OracleSettings()
        s2.samplingFreq = 16367600.0
        fo, po, ho = orc.probe_stats(s2, x[off:off + n])
        assert nseg == (n - 1024) // 15360
        assert np.array_equal(hist, ho) and np.array_equal(f, fo)
        assert np.max(np.abs(pxx - po) / po) < PSD_TOL
    with pytest.raises(ValueError):
        ctx.probe_stats(rec, 0, 16383, 38.192)
    with pytest.raises(ValueError):
        ctx.probe_stats(rec, 400000, 200000, 38.192)
    out2 = s.probeData(m.DeviceSignal(rec, 12345, 381920), device=0)
    assert np.array_equal(out2["timeData"], x[12346:12345 + 38192 // 50])
    rec.free()


# ---- settings off the defaults, random scenes -------------------------------------------------------------------

def _oracle_vs_gpu(m, s, os_, rec_host, ms, split_env=None):
    n = s.samplesPerCode
    a = m.AcquisitionResult(s, device=0)
    try:
        a.acquire(rec_host[:11 * n])
    except IndexError:
        # the reference's own failure (coarse code phase == samples per chip, acquisition.py:152-162; round 6: wide-fuzz
        # seed 1401, 5 samples per chip and a code phase of 5): the oracle must raise it too, and nothing is tracked
        with pytest.raises(IndexError):
            orc.acquire(os_, rec_host[:11 * n])
        return a, None
    ref = orc.acquire(os_, rec_host[:11 * n])
    assert np.array_equal(a.codePhase, ref["codePhase"])
    assert np.array_equal(a.carrFreq, ref["carrFreq"])
    assert np.allclose(a.peakMetric, ref["peakMetric"], rtol=1e-9, atol=0)
    if not np.any(a.carrFreq):
        return a, None
    a.preRun()
    chans_ref = orc.pre_run(os_, ref)
    assert np.array_equal(a.channels.PRN, chans_ref["PRN"])
    t = m.TrackingResult(a, device=0)
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload(rec_host)
    old = {k: os.environ.get(k) for k in (split_env or {})}
    os.environ.update(split_env or {})
    try:
        t.track(m.DeviceFile(rec))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        rec.free()
    want = orc.stack_series(orc.track(os_, chans_ref, rec_host))
    assert np.array_equal(t.series[:, 0], want[:, 0])
    assert _trk_err(t.series, want) < TRK_TOL
    return a, t


def test_correlator_spacing_outside_one_chip_is_refused(default_record):
    m = pkg()
    g = load_golden("trk_default.npz")
    for bad in (0.0, 1.0, 1.5, -0.25):
        s, t = _golden_tracker(m, g, ms=10)
        s.dllCorrelatorSpacing = bad
        t = m.TrackingResult(type("A", (), {"channels": t._channels, "settings": s})(), device=0)
        with pytest.raises(RuntimeError):
            t.track(m.DeviceFile(m.engine.get_context(s, 0).upload(default_record[:40 * 38192])))


@pytest.mark.parametrize("spacing,dll_bw,pll_bw,band,split", [(0.25, 1.0, 15.0, 10.0, None), (0.4, 4.0, 40.0, 6.0, "1"),
                                                              (0.1, 2.0, 25.0, 14.0, "3"), (0.8, 2.0, 25.0, 14.0, None)])
def test_non_default_loop_and_search_settings(spacing, dll_bw, pll_bw, band, split):
    """Correlator spacing, loop bandwidths, damping, search band and threshold off the defaults: the three code
    ramps then switch chips at different samples (exact per-sample path of the map)."""
    m = pkg()
    kw = dict(dllCorrelatorSpacing=spacing, dllNoiseBandwidth=dll_bw, pllNoiseBandwidth=pll_bw, acqSearchBand=band,
              acqThreshold=2.2, dllDampingRatio=0.8, pllDampingRatio=0.6, numberOfChannels=3, msToProcess=60.0,
              acqSatelliteList=range(1, 9))    # the reference searches PRN 1..len(list) (acquisition.py:103)
    s = m.Settings()
    os_ = orc.OracleSettings()
    for k, v in kw.items():
        setattr(s, k, v)
        setattr(os_, k, v)
    sc = m.synth.Scene.make(0x5E771 + int(spacing * 100), s.samplingFreq, s.IF, [3, 7, 8], [1200, -2100, 2600],
                            [100, 20000, 31000], [9, 8, 8])
    rec = m.synth.generate(sc, m.synth.record_length(s.samplesPerCode, 60))
    a, t = _oracle_vs_gpu(m, s, os_, rec, 60, {"SGX_TRK_SPLIT": split} if split else None)
    assert t is not None and set(a.channels.PRN[:3]) == {3, 7, 8}
    # the throughput-mode kernel (more than 128 channels) with the same settings: replicas of the three channels
    ctx = m.engine.get_context(s, 0)
    dev = ctx.upload(rec)
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    many, done = ctx.track(dev, [chans[i % 3] for i in range(132)], 60)
    dev.free()
    assert np.all(done == 60)
    for i in range(132):
        assert np.array_equal(many[i, 0], t.series[i % 3, 0])
        assert np.array_equal(many[i], many[i % 3])
    assert _trk_err(many[:3], t.series) < 1e-9


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scenes_against_oracle(seed):
    """Random Doppler / code phase / amplitude (down to signals near the detection threshold), random PRN subsets;
    one scene without any satellite."""
    m = pkg()
    rng = np.random.default_rng(seed)
    s = m.Settings()
    os_ = orc.OracleSettings()
    prns = sorted(rng.choice(np.arange(1, 7), size=3, replace=False).tolist())
    present = prns[:2] if seed != 13 else []
    for o in (s, os_):
        o.acqSatelliteList = range(1, 7)      # PRN 1..6 are searched (acquisition.py:103 loops over range(len(list)))
        o.numberOfChannels = 2
        o.msToProcess = 40.0
    n = s.samplesPerCode
    sc = m.synth.Scene.make(0xF00D0000 + seed, s.samplingFreq, s.IF, present,
                            [float(rng.integers(-6500, 6500)) for _ in present],
                            [int(rng.integers(0, n)) for _ in present],
                            [int(rng.integers(2, 10)) for _ in present])
    rec = m.synth.generate(sc, m.synth.record_length(n, 40))
    a, t = _oracle_vs_gpu(m, s, os_, rec, 40)
    if seed == 13:
        assert t is None and not np.any(a.carrFreq)


def test_decode_ephemerides_end_to_end():
    """32.1 s of a scene with decodable navigation frames: track -> findPreambles -> bits -> sgx_ephemeris; the
    decoded parameters and TOW must be the ones the generator transmitted (decoded from its bit table directly by
    the oracle, which is pinned on the reference's ephemeris())."""
    g = load_golden("nav_preambles.npz")
    m = pkg()
    s = _nav_settings(m)
    s.msToProcess = 32100.0
    ctx = m.engine.get_context(s, 0)
    sc = scene_from_json(g["scene"]).with_nav_message(first_boundary=100, tow0=4321, first_id=3)
    rec = ctx.synth(sc, m.synth.record_length(s.samplesPerCode, 32100))
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([g["ch_PRN"], g["ch_acquiredFreq"], g["ch_codePhase"], ['T', 'T']],
                                    names='PRN,acquiredFreq,codePhase,status')
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    rec.free()
    nav = m.NavigationResult(t, device=0)
    eph, tow, first, active = nav.decodeEphemerides()
    assert list(first) == [1999, 1999] and list(active) == [0, 1]
    assert tow == (4321 + 1 + 4) * 6 - 30
    for ch, prn in enumerate(g["ch_PRN"]):
        tab = sc.nav_bits[ch]
        want, tow_w = orc.ephemeris([str(int(b)) for b in tab[100:1600]], str(int(tab[99])))
        got = tuple(eph[int(prn) - 1])
        assert got == want and tow_w == tow
        assert isinstance(got[0], int) and isinstance(got[3], float)
    assert eph[0].IODC is None      # PRN 1 was not tracked


def test_position_fix_from_if_samples():
    """The whole receiver on a physically consistent scene (tests/nav_scene.py): 37 s of IF samples generated in HBM
    -> acquisition -> tracking -> preambles -> ephemerides -> pseudoranges -> least squares.  The fix must land
    on the receiver position the scene was built for, and the decoded ephemerides must be the transmitted ones."""
    import nav_scene
    m = pkg()
    sc, truth = nav_scene.build()
    s = m.Settings()
    nsat = len(truth["prns"])
    s.samplingFreq, s.IF, s.msToProcess, s.numberOfChannels = 16368000.0, 4130400.0, 37000.0, nsat
    ctx = m.engine.get_context(s, 0)
    n = s.samplesPerCode
    rec = ctx.synth(sc, m.synth.record_length(n, 37000))
    a = m.AcquisitionResult(s, device=0)
    a.acquire(m.DeviceSignal(rec, 0, 11 * n))
    a.preRun()
    assert sorted(int(p) for p in a.channels.PRN) == sorted(truth["prns"]) and truth["gdop"] < 4.0
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    rec.free()
    nav = m.NavigationResult(t, device=0)
    nav.postNavigate()
    sol = nav.solutions[0]
    n_meas = int((37000 - 5200) / 500)
    xyz = np.stack([sol.X, sol.Y, sol.Z])[:, :n_meas]
    assert np.all(np.isfinite(xyz)) and np.all(np.isnan(sol.X[n_meas:]))
    # truth.  The scene keeps every Doppler constant, so the modelled range is off by up to a*t^2/2 (tens of metres
    # after half a minute), and the reference's pseudoranges are whole samples (18 m here): the first fixes are
    # the sharp ones
    err = np.linalg.norm(xyz - truth["rx"][:, None], axis=0)
    assert err[:6].max() < 60.0 and np.median(err) < 100.0 and err.max() < 250.0, err
    assert abs(np.median(sol.latitude[:n_meas]) - truth["site"][0]) < 1e-3
    assert abs(np.median(sol.longitude[:n_meas]) - truth["site"][1]) < 1e-3
    assert sol.utmZone == 13 and np.all(sol.DOP[0, :n_meas] > 1.0) and np.all(sol.DOP[0, :n_meas] < 5.0)
    # parity: the oracle's restatement of the same chain on the same tracking output, several epochs
    so = orc.OracleSettings(samplingFreq=s.samplingFreq, IF=s.IF, numberOfChannels=nsat, msToProcess=37000.0)
    rows = [np.asarray(r.absoluteSample, dtype=np.float64) for r in t.results]
    ip = np.stack([np.asarray(r.I_P, dtype=np.float64) for r in t.results])
    first, active = orc.find_preambles(ip, ['T'] * nsat, nsat)
    assert np.array_equal(first, nav.findPreambles()[0]) and np.all(np.abs(first - 5210) <= 12)
    table = np.zeros((32, 27))
    tow = None
    for c_ in active:
        bits = [str(int(b)) for b in orc.nav_bits(ip[c_], int(first[c_]))]
        dec, tow = orc.ephemeris(bits[1:], bits[0])
        table[int(t.results[c_].PRN) - 1] = [float(v) for v in dec]
    assert tow == truth["tow"]
    prn_act = [int(t.results[c_].PRN) for c_ in active]
    for k in (0, 1, 17, 40, n_meas - 1):
        raw = orc.calculate_pseudoranges(so, rows, first + 500.0 * k, active)
        sat, clk = orc.satpos(tow + 0.5 * k, prn_act, table)
        p, el, az, dop = orc.least_square_pos(sat, raw[active] + clk * so.c, so.c, True)
        assert np.max(np.abs(p[:3] - xyz[:, k])) < 1e-5 and abs(p[3] - sol.dt[k]) < 1e-5
        assert np.max(np.abs(dop - sol.DOP[:, k])) < 1e-9
        ch = sol.channel[0]
        assert np.max(np.abs(ch.rawP[:, k] - raw)) == 0.0
        assert np.max(np.abs(ch.el[active, k] - el)) < 1e-9 and np.max(np.abs(ch.az[active, k] - az)) < 1e-9
        lat, lon, h = orc.cart2geo(p[0], p[1], p[2], 4)
        assert abs(lat - sol.latitude[k]) < 1e-10 and abs(lon - sol.longitude[k]) < 1e-10 and abs(h - sol.height[k]) < 1e-5
        E, N, U = orc.cart2utm(p[0], p[1], p[2], orc.find_utm_zone(lat, lon))
        assert abs(E - sol.E[k]) < 1e-5 and abs(N - sol.N[k]) < 1e-5 and abs(U - sol.U[k]) < 1e-5
    # the tracker against the generator: every code period starts within a sample of where the scene's code NCO
    # puts it (true arrival of the subframe + whole code periods at the satellite's code rate)
    for c_ in active:
        prn = int(t.results[c_].PRN)
        sat_ = [q for q in sc.sats if q["prn"] == prn][0]
        arrival = truth["arrival_samples"][truth["prns"].index(prn)]
        per = 1023.0 * 2 ** 32 / sat_["code_fcw"]                        # samples per code period
        kk = np.arange(0, 37000 - int(first[c_]))
        # absoluteSample is the file position AFTER the block (tracking.py:255): the next period's first sample,
        # i.e. the true boundary rounded up to a whole sample, plus the DLL's jitter
        d = rows[c_][int(first[c_]):] - (arrival + (kk + 1) * per)
        assert -0.35 < d.min() and d.max() < 1.35, (prn, d.min(), d.max())
    for prn in truth["prns"]:
        got = np.array([float(v) for v in nav.ephemeris[prn - 1]])
        assert np.array_equal(got, truth["eph_table"][prn - 1])
    ch = sol.channel[0]
    assert np.all(np.sort(ch.PRN[:, 0]) == np.sort(truth["prns"]))
    assert np.all(ch.el[:, 1] > 10.0)
    print("position error over %d fixes: median %.1f m, max %.1f m" % (n_meas, np.median(err), err.max()))
    # and the reference itself, run on the host-generated twin of this record (fixture fix_scene.npz: its own
    # acquisition, tracking and postNavigate): same channels, same block boundaries, same fixes
    from test_geo_functions import compare_solutions, rebuild_tracking
    g = load_golden("fix_scene.npz")
    assert [int(p) for p in t.results.PRN] == [int(p) for p in g["PRN"]]
    assert np.array_equal(a.channels.acquiredFreq[:nsat], g["ch_acquiredFreq"][:nsat])
    assert np.array_equal(a.channels.codePhase[:nsat], g["ch_codePhase"][:nsat])
    _, _, abs_ref, _ = rebuild_tracking(g)
    assert np.array_equal(np.stack(rows), np.stack(abs_ref))
    assert np.array_equal(np.packbits(ip > 0, axis=1), g["ip_sign"])
    assert np.max(np.abs(np.sqrt(np.mean(ip ** 2, axis=1)) / g["ip_rms"] - 1)) < 1e-9
    assert np.array_equal(np.asarray(first), g["firstSubFrame"])
    got = {k: sol[k] for k in ("X", "Y", "Z", "dt", "latitude", "longitude", "height", "E", "N", "U", "DOP", "utmZone")}
    got.update({k: sol.channel[0][k] for k in ("rawP", "correctedP", "el", "az")})
    compare_solutions(got, g, tol_m=1e-5)
    assert np.array_equal(sol.channel[0].PRN.astype(np.float64), g["chPRN"])


def test_post_navigate_on_reference_tracking_output():
    """NavigationResult.postNavigate on the tracking output of the reference's own tracker (rebuilt from the compact
    fixture) against the reference's postNavigate."""
    from test_geo_functions import compare_solutions, rebuild_tracking
    g = load_golden("fix_scene.npz")
    m = pkg()
    prn, status, abs_rows, ip_rows = rebuild_tracking(g)
    s = m.Settings()
    s.samplingFreq, s.IF, s.msToProcess, s.numberOfChannels = 16368000.0, 4130400.0, 37000.0, len(prn)

    class Trk(object):
        pass

    t = Trk()
    t.settings, t.channels = s, None
    t.results = np.recarray((len(prn),), dtype=[('status', 'S1'), ('absoluteSample', 'O'), ('I_P', 'O'), ('PRN', 'i8')])
    for i in range(len(prn)):
        t.results[i].status, t.results[i].PRN = b'T', prn[i]
        t.results[i].absoluteSample, t.results[i].I_P = abs_rows[i], ip_rows[i]
    nav = m.NavigationResult(t, device=0)
    nav.postNavigate()
    sol = nav.solutions[0]
    got = {k: sol[k] for k in ("X", "Y", "Z", "dt", "latitude", "longitude", "height", "E", "N", "U", "DOP", "utmZone")}
    got.update({k: sol.channel[0][k] for k in ("rawP", "correctedP", "el", "az")})
    compare_solutions(got, g, tol_m=1e-5)
    tab = np.zeros((32, 27))
    for i in range(32):
        if nav.ephemeris[i].IODC is not None:
            tab[i] = [float(v) for v in nav.ephemeris[i]]
    assert np.array_equal(tab, g["eph"])
    # too short a record / too few channels: the reference prints a message and leaves the results unset
    s.msToProcess = 35000.0
    nav2 = m.NavigationResult(t, device=0)
    nav2.postNavigate()
    assert nav2._solutions is None and nav2._eph is None


def test_post_processing_from_a_record_file(tmp_path):
    """Settings.postProcessing(fileName) - the reference's top-level call (initialize.py:420-515) - on a 606 MB record
    file: streamed into HBM, acquired, tracked and navigated; same fixes as the reference's own run (fix_scene.npz)."""
    import nav_scene
    from test_geo_functions import compare_solutions
    g = load_golden("fix_scene.npz")
    m = pkg()
    sc, truth = nav_scene.build()
    s = m.Settings()
    s.samplingFreq, s.IF, s.msToProcess, s.numberOfChannels = 16368000.0, 4130400.0, 37000.0, len(truth["prns"])
    ctx = m.engine.get_context(s, 0)
    rec = ctx.synth(sc, m.synth.record_length(s.samplesPerCode, 37000))
    path = str(tmp_path / "record.bin")
    rec.download().tofile(path)
    rec.free()
    acq, trk, nav = s.postProcessing(path)
    sol = nav.solutions[0]
    got = {k: sol[k] for k in ("X", "Y", "Z", "dt", "latitude", "longitude", "height", "E", "N", "U", "DOP", "utmZone")}
    got.update({k: sol.channel[0][k] for k in ("rawP", "correctedP", "el", "az")})
    compare_solutions(got, g, tol_m=1e-5)
    assert [int(p) for p in trk.results.PRN] == [int(p) for p in g["PRN"]]
    # nothing to acquire: the reference prints a message and stops
    quiet = str(tmp_path / "noise.bin")
    m.synth.generate(m.synth.Scene.make(5, s.samplingFreq, s.IF, [], [], [], []), 12 * s.samplesPerCode).tofile(quiet)
    s2 = m.Settings()
    s2.samplingFreq, s2.IF = s.samplingFreq, s.IF
    a2, t2, n2 = s2.postProcessing(quiet)
    assert t2 is None and n2 is None and not np.any(a2.carrFreq)


def test_streaming_record_overlaps_tracking_with_identical_results(full_run, tmp_path):
    """sgx_if_open_file: the record fills in the background while the cooperative kernel follows the watermark."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    path = str(tmp_path / "stream.bin")
    n_bytes = m.synth.record_length(s.samplesPerCode, 6000)
    rec.download(0, n_bytes).tofile(path)
    for env in ({}, {"SGX_TRK_STREAM": "0"}, {"SGX_TRK_V3": "0"}, {"SGX_TRK_SPLIT": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = ctx.open_file(path, 0, n_bytes)
            s2, d2 = ctx.track(r, chans, 6000)           # starts while the file is still being read
            assert np.all(d2 == 6000) and np.array_equal(s2[:, 0], series[:, 0, :6000])
            tm = ctx.timing()
            if "SGX_TRK_V3" in env:
                # the round-3 kernel (30 members per channel) following the watermark: another reduction order
                assert tm["track_kernel"] == 2 and tm["track_members"] == 30 and tm["track_streamed"] == 1
                assert _trk_err(s2, series[:, :, :6000]) < 1e-9
            elif "SGX_TRK_SPLIT" not in env:
                # the resident run's kernel (its record wave follows the watermark, or - SGX_TRK_STREAM=0 - the host
                # waits for the whole record first): bit-identical
                assert tm["track_kernel"] == 5 and tm["track_members"] == 20
                assert tm["track_streamed"] == (0 if env else 1)
                assert np.array_equal(s2, series[:, :, :6000])
            else:
                # one workgroup per channel owning all ten units, following the watermark too: another reduction
                # order, equal to rounding
                assert tm["track_kernel"] == 2 and tm["track_members"] == 1 and tm["track_streamed"] == 1
                assert _trk_err(s2, series[:, :, :6000]) < 1e-9
            assert np.array_equal(r.download(n_bytes - 5000, 5000), rec.download(n_bytes - 5000, 5000))
            r.free()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    # acquisition and statistics on a record that is still streaming wait for their window only
    r = ctx.open_file(path, 0, n_bytes)
    a2 = m.AcquisitionResult(s, device=0)
    a2.acquire(m.DeviceSignal(r, 0, 11 * s.samplesPerCode))
    assert np.array_equal(a2.carrFreq, a.carrFreq) and np.array_equal(a2.codePhase, a.codePhase)
    r.wait()
    r.free()
    # an unaligned window of the file, freed while the transfer is still running
    r = ctx.open_file(path, 12345, 40 * 1000 * 1000)
    assert len(r) == 40 * 1000 * 1000
    r.free()
    # the tracker's short-read behaviour on a streaming record
    r = ctx.open_file(path, 0, 50 * s.samplesPerCode)
    s3, d3 = ctx.track(r, chans, 100)
    assert np.all(d3 < 100) and np.all(d3 >= 47)
    r.free()
    with pytest.raises(RuntimeError):
        ctx.open_file(str(tmp_path / "missing.bin"), 0, 10)


@pytest.mark.slow
def test_full_config3_run_against_the_oracle(full_run):
    """BASELINE config 3 at its full size against the oracle itself (not only through properties): 8 channels x
    37 000 ms = 296 000 dependent blocks; the numpy restatement runs in eight host processes (about 40 s)."""
    from concurrent.futures import ProcessPoolExecutor
    from oracle_helpers import oracle_channel
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    host = rec.download()
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        want = np.stack(list(ex.map(oracle_channel, [(host, p, f, c, 37000) for p, f, c in chans])))
    assert np.array_equal(series[:, 0], want[:, 0])                  # every block boundary of the run
    assert _trk_err(series, want) < 1e-9                              # bar: 1e-6
    assert np.max(np.abs(series[:, 1] - want[:, 1])) < 1e-8 and np.max(np.abs(series[:, 2] - want[:, 2])) < 1e-7


@pytest.mark.slow
def test_config4_full_noncoherent_acquisition_against_the_oracle(default_record):
    """BASELINE config 4 at its full size: 32 PRNs, 10 x 1 ms non-coherent, 29 bins (9 280 correlations); the
    oracle takes about half a minute of host time."""
    m = pkg()
    s = m.Settings()
    n = s.samplesPerCode
    a = m.AcquisitionResult(s, device=0)
    a.acquire(default_record[:20 * n], n_blocks=10, noncoh=True)
    r = orc.acquire(orc.OracleSettings(), default_record[:20 * n], n_blocks=10, noncoh=True)
    assert np.array_equal(a.codePhase, r["codePhase"]) and np.array_equal(a.carrFreq, r["carrFreq"])
    assert np.array_equal(a.internals["freqBin"], r["freqBin"])
    assert np.allclose(a.peakMetric, r["peakMetric"], rtol=1e-9, atol=0)
    assert np.count_nonzero(a.carrFreq) == 8


def test_acquisition_chunks_and_queues_do_not_change_the_result(default_record):
    """The correlation batch is cut into PRN chunks (SGX_ACQ_CHUNK_ROWS rows each) that alternate between two queues
    (SGX_ACQ_STREAMS); non-coherent sums whose PRN does not fit half a chunk go in runs of Doppler bins.  Every cut gives the
    same bits (a wrong output offset of the bin runs with more than one PRN per chunk went unnoticed in round 5 until a
    chunk size off the default was tried)."""
    m = pkg()
    s = m.Settings()
    n = s.samplesPerCode
    x = default_record[:20 * n]

    def run(env):
        os.environ.update(env)
        try:
            a2 = m.AcquisitionResult(s, device=0)
            a2.acquire(x[:11 * n])
            a4 = m.AcquisitionResult(s, device=0)
            a4.acquire(x, n_blocks=10, noncoh=True, prn_indices=list(range(12)))
        finally:
            for k in env:
                os.environ.pop(k, None)
        return a2, a4

    r2, r4 = run({"SGX_ACQ_STREAMS": "1"})
    assert np.count_nonzero(r2.carrFreq) == 8
    for env in ({}, {"SGX_ACQ_CHUNK_ROWS": "120"}, {"SGX_ACQ_CHUNK_ROWS": "200"}, {"SGX_ACQ_CHUNK_ROWS": "580"},
                {"SGX_ACQ_CHUNK_ROWS": "1160"}, {"SGX_ACQ_CHUNK_ROWS": "1160", "SGX_ACQ_STREAMS": "1"},
                {"SGX_ACQ_CHUNK_ROWS": "60"}):
        a2, a4 = run(dict(env))
        for a, r in ((a2, r2), (a4, r4)):
            assert np.array_equal(a.peakMetric, r.peakMetric), env
            assert np.array_equal(a.codePhase, r.codePhase) and np.array_equal(a.carrFreq, r.carrFreq), env
            assert np.array_equal(a.internals["freqBin"], r.internals["freqBin"]), env


def test_config5_full_64_replicated_channels(full_run):
    """BASELINE config 5 at its full size on one GPU: 64 channels (8 initialisations x 8 replicas) x 37 000 ms.
    Replicas are bit-identical; block boundaries equal the 8-channel run's, sums agree to rounding (the launch
    uses 4 cooperating workgroups per channel instead of 10, i.e. another summation order)."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    many = [chans[i % 8] for i in range(64)]
    s64, d64 = ctx.track(rec, many, 37000)
    tm = ctx.timing()
    assert tm["track_kernel"] == 2 and tm["track_members"] == 4      # four members per channel, three / three / two / two units
    assert np.all(d64 == 37000)
    for i in range(8, 64):
        assert np.array_equal(s64[i], s64[i % 8])
    assert np.array_equal(s64[:8, 0], series[:, 0])
    assert _trk_err(s64[:8], series) < 1e-9


def test_independent_records_on_one_gpu_at_once(default_record):
    """Three threads, each with a private context (stream, scratch, record), acquire and track at the same time:
    every one of them reproduces the reference's golden output, whatever share of the CUs its launch got."""
    import threading
    g = load_golden("trk_default.npz")
    ga = load_golden("acq_default.npz")
    m = pkg()
    out, errs = {}, []

    def worker(k):
        try:
            s = m.Settings()
            s.numberOfChannels = 4
            s.msToProcess = 400.0
            with m.engine.private_context(s, 0, priority=(-1, 1, 0)[k]) as ctx:
                rec = ctx.upload(default_record)
                a = m.AcquisitionResult(s, device=0)
                a.acquire(m.DeviceSignal(rec, 0, 11 * 38192))
                a.preRun()
                t = m.TrackingResult(a, device=0)
                t.track(m.DeviceFile(rec))
                out[k] = (a.carrFreq.copy(), a.codePhase.copy(), t.series.copy())
                rec.free()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in range(3):
        cf, cp, series = out[k]
        assert np.array_equal(cf, ga["carrFreq"]) and np.array_equal(cp, ga["codePhase"])
        assert np.array_equal(series[:, 0], g["series"][:, 0]) and _trk_err(series, g["series"]) < TRK_TOL
        # (which member layout a launch gets depends on the CUs the other two hold at that moment: same block
        # boundaries, sums equal to rounding)
        assert np.array_equal(series[:, 0], out[0][2][:, 0]) and _trk_err(series, out[0][2]) < 1e-9


def test_streaming_record_with_many_live_streams(full_run, tmp_path, capfd):
    """HIP maps streams onto a few hardware queues.  The copy stream of a streaming record has the highest
    priority (its own queue pool), so it runs beside the tracking kernel however many streams are alive; with a
    normal-priority copy stream (test hook) the watermark can stall behind the kernel - then the kernel gives up
    after about a second and the launch is repeated on the resident record.  Same block boundaries, sums equal to
    rounding in every case."""
    m, s, ctx, sc, rec, a, chans, series, done = full_run
    path = str(tmp_path / "stream.bin")
    ms = 3000
    n_bytes = m.synth.record_length(s.samplesPerCode, ms)
    rec.download(0, n_bytes).tofile(path)
    extra = []
    for k in range(13):
        s2 = m.Settings()
        s2.acqThreshold = 2.5 + 0.01 * (k + 1)
        extra.append(m._native.Context(s2, 0))
    try:
        for _ in range(4):
            r = ctx.open_file(path, 0, n_bytes)
            s2_, d2 = ctx.track(r, chans, ms)
            r.free()
            # (the watermark is followed by the resident run's kernel: bit-identical)
            assert np.all(d2 == ms) and np.array_equal(s2_, series[:, :, :ms])
        assert "did not stream in" not in capfd.readouterr().err
        os.environ["SGX_STREAM_PRIO"] = "0"
        try:
            for _ in range(3):
                r = ctx.open_file(path, 0, n_bytes)
                s3, d3 = ctx.track(r, chans, ms)
                r.free()
                assert np.all(d3 == ms) and np.array_equal(s3, series[:, :, :ms])
        finally:
            os.environ.pop("SGX_STREAM_PRIO", None)
    finally:
        for c in extra:
            c.close()


@pytest.mark.parametrize("fs,IF", [(5456000.0, 1364000.0), (61380000.0, 15345000.0), (4092000.0, 1023000.0)])
def test_other_front_ends_against_oracle(fs, IF):
    """Sampling rates far from the default: samplesPerCode 5456 (two tracking units), 61 380 = 2^2 3^2 5 11 31
    (fifteen units, other FFT radices) and 4092 (one unit, exactly four samples per chip)."""
    m = pkg()
    s = m.Settings()
    os_ = orc.OracleSettings()
    for o in (s, os_):
        o.samplingFreq, o.IF = fs, IF
        o.acqSatelliteList = range(1, 7)
        o.numberOfChannels = 2
        o.msToProcess = 50.0
    n = s.samplesPerCode
    assert n == int(round(fs / 1000))
    sc = m.synth.Scene.make(0xFE000 + n, fs, IF, [2, 5], [1750.0, -3300.0], [n // 3, n - 5], [9, 8])
    rec = m.synth.generate(sc, m.synth.record_length(n, 50))
    a, t = _oracle_vs_gpu(m, s, os_, rec, 50)
    assert t is not None and sorted(int(p) for p in a.channels.PRN) == [2, 5]
    # the other kernels on the same front end: one workgroup per channel, and the throughput-mode kernel
    ctx = m.engine.get_context(s, 0)
    dev = ctx.upload(rec)
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
    os.environ["SGX_TRK_SPLIT"] = "1"
    try:
        one, d1 = ctx.track(dev, chans, 50)
    finally:
        os.environ.pop("SGX_TRK_SPLIT", None)
    many, dm = ctx.track(dev, [chans[i % 2] for i in range(130)], 50)
    dev.free()
    assert np.all(d1 == 50) and np.all(dm == 50)
    assert np.array_equal(one[:, 0], t.series[:, 0]) and _trk_err(one, t.series) < 1e-9
    assert np.array_equal(many[:2, 0], t.series[:, 0]) and _trk_err(many[:2], t.series) < 1e-9
    for i in range(2, 130):
        assert np.array_equal(many[i], many[i % 2])


def test_speculative_kernel_is_for_half_chip_spacing_only():
    """60 Msps with dllCorrelatorSpacing 0.32: the gaps between the arms' chip boundaries (0.32 chips = 18.8 samples) pass
    the 18-sample test of the speculative kernel, whose fused ramp is cut for a spacing of exactly half a chip - the host
    must send this to the round-3 kernel (track_kernel 2), and the result must be the oracle's."""
    m = pkg()
    s = m.Settings()
    os_ = orc.OracleSettings()
    fs, IF = 60000000.0, 15000000.0
    for o in (s, os_):
        o.samplingFreq, o.IF = fs, IF
        o.dllCorrelatorSpacing = 0.32
        o.acqSatelliteList = range(1, 7)
        o.numberOfChannels = 2
        o.msToProcess = 60.0
    n = s.samplesPerCode
    sc = m.synth.Scene.make(0xFE000 + n, fs, IF, [2, 5], [1750.0, -3300.0], [n // 3, n - 5], [9, 8])
    rec = m.synth.generate(sc, m.synth.record_length(n, 60))
    a, t = _oracle_vs_gpu(m, s, os_, rec, 60)
    assert t is not None
    assert int(m.engine.get_context(s, 0).timing()["track_kernel"]) == 2
    # the same front end at half a chip does take the speculative kernel, against the oracle as well
    for o in (s, os_):
        o.dllCorrelatorSpacing = 0.5
    a, t = _oracle_vs_gpu(m, s, os_, rec, 60)
    assert t is not None
    assert int(m.engine.get_context(s, 0).timing()["track_kernel"]) == 5


def test_tiny_runs_and_unsupported_sample_type(default_record):
    """msToProcess of 1, 2 and 3 code periods (first block, first filter update) and a non-int8 dataType."""
    g = load_golden("trk_default.npz")
    m = pkg()
    for ms in (1, 2, 3):
        s, t = _golden_tracker(m, g, ms=ms)
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            default_record[:6 * 38192].tofile(f.name)
            with open(f.name, "rb") as fid:
                t.track(fid)
        assert t.series.shape == (4, 13, ms)
        assert np.array_equal(t.series[:, 0], g["series"][:, 0, :ms])
        assert _trk_err(t.series, g["series"][:, :, :ms]) < TRK_TOL
    s, t = _golden_tracker(m, g, ms=5)
    s.dataType = 'complex64'
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        default_record[:8 * 38192].tofile(f.name)
        with open(f.name, "rb") as fid:
            with pytest.raises(TypeError):
                t.track(fid)


def _int16_tracker(m, g, case, ms=None, prn=None, freq=None, phase=None, skip=None):
    s = m.Settings()
    s.dataType = 'int16'
    prn = g[case + "_PRN"] if prn is None else prn
    s.numberOfChannels = len(prn)
    s.msToProcess = float(int(g["ms"]) if ms is None else ms)
    s.skipNumberOfBytes = int(g[case + "_skip"]) if skip is None else skip
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([prn, g[case + "_acquiredFreq"] if freq is None else freq,
                                     g[case + "_codePhase"] if phase is None else phase, ['T'] * len(prn)],
                                    names='PRN,acquiredFreq,codePhase,status')
    return s, m.TrackingResult(a, device=0)


def test_track_int16_record_matches_reference():
    """Settings.dataType = 'int16' (tracking.py:154) with the reference's byte seeks and byte positions
    (tracking.py:107, 255): its own outputs (tests/golden/trk_int16.npz) for a locked channel and for two channels
    started where the byte seek lands; through a real file and through a record already in HBM."""
    g = load_golden("trk_int16.npz")
    m = pkg()
    rec16 = (m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"])).astype(np.int16)
             * int(g["scale"])).astype("<i2")
    # the reference's acquisition of the same int16 array (acquisition.py:55-59 takes whatever dtype it is handed)
    sa = m.Settings()
    sa.dataType = 'int16'
    acq = m.AcquisitionResult(sa, device=0)
    acq.acquire(rec16[:11 * 38192])
    assert np.array_equal(acq.codePhase, g["codePhase"]) and np.array_equal(acq.carrFreq, g["carrFreq"])
    assert np.allclose(acq.peakMetric, g["peakMetric"], rtol=1e-9, atol=0)
    for case in ("locked", "as_is"):
        want = g[case + "_series"]
        s, t = _int16_tracker(m, g, case)
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            rec16.tofile(f.name)
            with open(f.name, "rb") as fid:
                t.track(fid)
                assert fid.tell() == int(want[-1, 0, -1])
        assert t.series.shape == want.shape
        assert np.array_equal(t.series[:, 0], want[:, 0]), case          # absoluteSample (bytes) bit-exact
        assert _trk_err(t.series, want) < TRK_TOL, case
        assert np.max(np.abs(t.series[:, 1] - want[:, 1])) < 1e-6
        assert np.max(np.abs(t.series[:, 2] - want[:, 2])) < 1e-5
        # the same from HBM
        ctx = m.engine.get_context(s, 0)
        dev = m.DeviceFile(ctx.upload_bytes(rec16), 0)
        s2, t2 = _int16_tracker(m, g, case)
        t2.track(dev)
        # (another alignment of the blocks in the record: another summation order, equal to rounding)
        assert np.array_equal(t2.series[:, 0], t.series[:, 0]) and _trk_err(t2.series, t.series) < 1e-9


def test_track_int16_every_member_layout_and_a_known_type():
    """int16 records run every member layout of the latency-mode kernel (one workgroup per channel, three members with
    several units, one per unit, one per unit and arm) with equal results, and more channels than the throughput-mode
    kernel's threshold (it reads int8 only: the latency-mode kernel takes them, one workgroup per channel); a sample type
    the C-ABI does not know is an argument error with a message, not a wrong result."""
    g = load_golden("trk_int16.npz")
    m = pkg()
    rec16 = (m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"])).astype(np.int16)
             * int(g["scale"])).astype("<i2")
    s, t = _int16_tracker(m, g, "locked", ms=5)
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload_bytes(rec16)
    chans = [(int(g["locked_PRN"][0]), float(g["locked_acquiredFreq"][0]), float(g["locked_codePhase"][0]))]
    with pytest.raises(m._native.SgxError, match="data_type"):
        ctx.track(rec, chans, 5, data_type=77)
    ser, done = ctx.track(rec, chans, 5, rec_file_offset=0, data_type=m._native.DT_INT16)
    assert ser.shape == (1, 13, 5) and ctx.timing()["track_members"] == 30
    for env, members in (({"SGX_TRK_SPLIT": "1"}, 1), ({"SGX_TRK_SPLIT": "3"}, 3), ({"SGX_TRK_ARMS": "3"}, 10)):
        os.environ.update(env)
        try:
            s2, d2 = ctx.track(rec, chans, 5, data_type=m._native.DT_INT16)
        finally:
            for k in env:
                os.environ.pop(k)
        assert ctx.timing()["track_members"] == members and np.all(d2 == 5)
        assert np.array_equal(s2[:, 0], ser[:, 0]) and _trk_err(s2, ser) < 1e-9
    many, dm = ctx.track(rec, chans * 130, 5, data_type=m._native.DT_INT16)
    tm = ctx.timing()
    assert tm["track_kernel"] == 3 and tm["track_members"] == 1 and np.all(dm == 5)
    assert all(np.array_equal(many[i], many[0]) for i in range(1, 130))
    assert np.array_equal(many[0, 0], ser[0, 0]) and _trk_err(many[:1], ser) < 1e-9


def test_track_int16_odd_start_byte_and_short_record():
    """A channel whose start byte skipNumberOfBytes + codePhase is odd reads int16 values that straddle the file's
    samples (the reference's fid.seek takes bytes, tracking.py:107) - followed as is, against the oracle; and the
    short-read exit (tracking.py:159-163) counted in bytes."""
    g = load_golden("trk_int16.npz")
    m = pkg()
    rec16 = (m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"])).astype(np.int16)
             * int(g["scale"])).astype("<i2")
    prn = np.array([int(g["locked_PRN"][0])] * 3)
    freq = np.array([float(g["locked_acquiredFreq"][0])] * 3)
    phase = np.array([12345.0, 2 * 12346.0, 7.0])        # odd byte, the sample-aligned start (locks), odd byte
    ms = 40
    s, t = _int16_tracker(m, g, "locked", ms=ms, prn=prn, freq=freq, phase=phase, skip=0)
    so = orc.OracleSettings(numberOfChannels=3, msToProcess=float(ms), dataType='int16', skipNumberOfBytes=0)
    want = orc.stack_series(orc.track(so, dict(PRN=prn, acquiredFreq=freq, codePhase=phase, status=['T'] * 3), rec16))
    dev = m.DeviceFile(m.engine.get_context(s, 0).upload_bytes(rec16), 0)
    t.track(dev)
    assert np.array_equal(t.series[:, 0], want[:, 0])
    assert _trk_err(t.series, want) < TRK_TOL
    assert np.sqrt(np.mean(want[1, 3] ** 2)) > 2 * np.sqrt(np.mean(want[0, 3] ** 2))    # only the aligned one locks
    # short record: one byte fewer than the last block of the last channel needs -> the reference's exit
    end = int(want[:, 0, -1].max())
    for cut, ok in ((end, True), (end - 1, False)):
        s3, t3 = _int16_tracker(m, g, "locked", ms=ms, prn=prn, freq=freq, phase=phase, skip=0)
        raw = rec16.view(np.int8)[:cut]
        dev = m.DeviceFile(m.engine.get_context(s3, 0).upload_bytes(raw), 0)
        t3.track(dev)
        assert (t3.series is not None) == ok
        o = orc.track(so, dict(PRN=prn, acquiredFreq=freq, codePhase=phase, status=['T'] * 3), raw)
        assert (o is not None) == ok
        # the same in throughput mode (150 channels: one workgroup each, byte planes of the int16 samples)
        ctx3 = m.engine.get_context(s3, 0)
        chans = [(int(prn[i]), float(freq[i]), float(phase[i])) for i in range(3)] * 50
        many, dm = ctx3.track(dev.record, chans, ms, data_type=m._native.DT_INT16)
        assert ctx3.timing()["track_kernel"] == 3
        if ok:
            assert np.all(dm == ms) and np.array_equal(many[:3, 0], want[:, 0]) and _trk_err(many[:3], want) < TRK_TOL
            assert all(np.array_equal(many[i], many[i % 3]) for i in range(3, 150))
        else:
            assert np.any(dm < ms)


def test_track_uint8_record_against_the_oracle(tmp_path):
    """Settings.dataType = 'uint8' (tracking.py:154 reads whatever numpy dtype the settings name): offset-binary bytes
    tracked as they are - the reference removes no offset - against the oracle on the same bytes: every member layout of
    the latency-mode kernel, through a real file (streamed) and from HBM; more channels than the throughput-mode kernel's
    threshold stay on the latency-mode kernel (one workgroup per channel)."""
    m = pkg()
    ms = 60
    s = m.Settings()
    s.dataType = 'uint8'
    s.numberOfChannels = 3
    s.msToProcess = float(ms)
    n = s.samplesPerCode
    sc = m.synth.Scene.default()
    rec8 = m.synth.generate(sc, m.synth.record_length(n, ms))
    recu = (rec8.astype(np.int16) + 128).astype(np.uint8)
    a8 = orc.acquire(orc.OracleSettings(), rec8[:11 * n])
    ch = orc.pre_run(orc.OracleSettings(numberOfChannels=3), a8)
    so = orc.OracleSettings(numberOfChannels=3, msToProcess=float(ms), dataType='uint8')
    want = orc.stack_series(orc.track(so, ch, recu))
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([ch["PRN"], ch["acquiredFreq"], ch["codePhase"], ['T'] * 3],
                                    names='PRN,acquiredFreq,codePhase,status')
    path = str(tmp_path / "u8.bin")
    recu.tofile(path)
    t = m.TrackingResult(a, device=0)
    with open(path, "rb") as fid:
        t.track(fid)
    assert np.array_equal(t.series[:, 0], want[:, 0]) and _trk_err(t.series, want) < TRK_TOL
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload_bytes(recu)
    chans = [(int(ch["PRN"][i]), float(ch["acquiredFreq"][i]), float(ch["codePhase"][i])) for i in range(3)]
    for env, members in (({}, 20), ({"SGX_TRK_V3": "0"}, 30), ({"SGX_TRK_SPLIT": "1"}, 1), ({"SGX_TRK_SPLIT": "3"}, 3),
                         ({"SGX_TRK_ARMS": "3"}, 10)):
        os.environ.update(env)
        try:
            s2, d2 = ctx.track(rec, chans, ms, data_type=m._native.DT_UINT8)
        finally:
            for k in env:
                os.environ.pop(k)
        assert ctx.timing()["track_members"] == members and np.all(d2 == ms)
        assert np.array_equal(s2[:, 0], want[:, 0]) and _trk_err(s2, want) < TRK_TOL
    many, dm = ctx.track(rec, chans * 50, 20, data_type=m._native.DT_UINT8)
    tm = ctx.timing()
    assert tm["track_kernel"] == 3 and tm["track_members"] == 1 and np.all(dm == 20)
    assert np.array_equal(many[:3, 0], want[:, 0, :20]) and _trk_err(many[:3], want[:, :, :20]) < TRK_TOL
    s.dataType = 'complex64'
    with pytest.raises(TypeError, match="little-endian real IF samples"):
        m.TrackingResult(a, device=0).track(open(path, "rb"))


def test_track_float32_record_matches_reference(tmp_path):
    """The reference's own outputs for a float32 file (tests/golden/trk_float32.npz; made by tests/golden/make_golden.py):
    block boundaries in bytes of the float file exactly, the series to rounding."""
    g = load_golden("trk_float32.npz")
    m = pkg()
    rec8 = m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]))
    recf = ((rec8.astype(np.int32) * 200 + 7) / 32768.0).astype("<f4")
    raw = np.concatenate([np.zeros(int(g["skip"]) // 4, "<f4"), recf])
    s = m.Settings()
    s.dataType = 'float32'
    s.numberOfChannels, s.msToProcess, s.skipNumberOfBytes = len(g["PRN"]), float(g["ms"]), int(g["skip"])
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([g["PRN"], g["acquiredFreq"], g["codePhase"], ['T'] * len(g["PRN"])],
                                    names='PRN,acquiredFreq,codePhase,status')
    path = str(tmp_path / "ref.f32")
    raw.tofile(path)
    t = m.TrackingResult(a, device=0)
    with open(path, "rb") as fid:
        t.track(fid)
    assert np.array_equal(t.series[:, 0], g["series"][:, 0]) and _trk_err(t.series, g["series"]) < TRK_TOL


def test_track_float32_record_by_exact_narrowing(tmp_path):
    """Settings.dataType = 'float32' (tracking.py:154): a record of floats that are integers times one power of two -
    written from ADC samples, or normalised by 2^15 - is tracked through the int8 / int16 kernels and scaled back, which
    is exact: against the oracle on the float bytes (positions in bytes of the float file, tracking.py:107, 255), and
    against the integer record's own run.  (Arbitrary floats and channels that start inside a sample take the
    per-sample kernel: test_track_any_sample_type_against_the_oracle.)"""
    m = pkg()
    ms = 40
    s = m.Settings()
    s.dataType = 'float32'
    s.numberOfChannels = 3
    s.msToProcess = float(ms)
    n = s.samplesPerCode
    rec8 = m.synth.generate(m.synth.Scene.default(), m.synth.record_length(n, ms))
    a8 = orc.acquire(orc.OracleSettings(), rec8[:11 * n])
    ch = orc.pre_run(orc.OracleSettings(numberOfChannels=3), a8)
    # the reference seeks BYTES: a float channel starts on a sample of the file when its start byte is a multiple of four
    skip = 4000
    phase = (np.asarray(ch["codePhase"], dtype=np.int64) * 4 - 4).astype(np.float64)   # sample codePhase - 1 -> byte 4 (codePhase - 1)
    ctx = m.engine.get_context(s, 0)
    cases = {"adc": rec8.astype(np.float32),                                         # k = 0, eight bits
             "normalised": ((rec8.astype(np.int32) * 200 + 7) / 32768.0).astype(np.float32)}   # k = 15, sixteen bits
    for name, recf in cases.items():
        raw = np.concatenate([np.zeros(skip // 4, np.float32), recf])
        so = orc.OracleSettings(numberOfChannels=3, msToProcess=float(ms), dataType='float32', skipNumberOfBytes=skip)
        want = orc.stack_series(orc.track(so, dict(PRN=ch["PRN"], acquiredFreq=ch["acquiredFreq"], codePhase=phase,
                                                   status=['T'] * 3), raw))
        s.skipNumberOfBytes = skip
        a = m.AcquisitionResult(s, device=0)
        a._channels = np.rec.fromarrays([ch["PRN"], ch["acquiredFreq"], phase, ['T'] * 3],
                                        names='PRN,acquiredFreq,codePhase,status')
        path = str(tmp_path / (name + ".f32"))
        raw.tofile(path)
        t = m.TrackingResult(a, device=0)
        with open(path, "rb") as fid:
            t.track(fid)                                   # a real file: streamed, then scanned and narrowed
            assert fid.tell() == int(want[-1, 0, ms - 1])
        assert np.array_equal(t.series[:, 0], want[:, 0]) and _trk_err(t.series, want) < TRK_TOL, name
        t2 = m.TrackingResult(a, device=0)
        t2.track(m.DeviceFile(ctx.upload_bytes(raw.view(np.int8)), 0))
        assert np.array_equal(t2.series, t.series)
        # the same floats without narrowing: the latency-mode kernel's float instance
        os.environ["SGX_TRK_F32_NARROW"] = "0"
        try:
            t3 = m.TrackingResult(a, device=0)
            t3.track(m.DeviceFile(ctx.upload_bytes(raw.view(np.int8)), 0))
        finally:
            os.environ.pop("SGX_TRK_F32_NARROW")
        assert m.engine.get_context(s, 0).timing()["track_members"] == 10
        assert np.array_equal(t3.series[:, 0], want[:, 0]) and _trk_err(t3.series, want) < TRK_TOL, name
    # the integers themselves through the int8 kernel: the float run is that run, positions in float bytes
    s8 = m.Settings()
    s8.numberOfChannels, s8.msToProcess, s8.skipNumberOfBytes = 3, float(ms), 0
    a8g = m.AcquisitionResult(s8, device=0)
    a8g._channels = np.rec.fromarrays([ch["PRN"], ch["acquiredFreq"], np.asarray(ch["codePhase"], dtype=np.float64) - 1.0, ['T'] * 3],
                                      names='PRN,acquiredFreq,codePhase,status')
    t8 = m.TrackingResult(a8g, device=0)
    t8.track(m.DeviceFile(m.engine.get_context(s8, 0).upload(rec8), 0))
    raw = np.concatenate([np.zeros(skip // 4, np.float32), cases["adc"]])
    tf = m.TrackingResult(a, device=0)
    tf.track(m.DeviceFile(ctx.upload_bytes(raw.view(np.int8)), 0))
    # (the narrowed record starts at the first channel's first sample: other 16-sample groups, another order of the
    # same additions)
    assert np.array_equal(tf.series[:, 0], t8.series[:, 0] * 4 + skip)
    same = t8.series.copy()
    same[:, 0] = tf.series[:, 0]
    assert _trk_err(tf.series, same) < 1e-10
    assert ctx.timing()["track_kernel"] in (2, 5)          # (the typed kernels ran)


def test_track_any_sample_type_against_the_oracle(tmp_path):
    """Settings.dataType is whatever numpy dtype the settings name (tracking.py:154): records of arbitrary float32 and
    float64 values on the latency-mode kernel (scaled by a power of two on conversion, sgx_trk2.hip) and - float16, the
    wider integers, or SGX_TRK_FLOAT_TYPED=0 - read sample by sample where they lie (sgx_trk_any.hip), promoted to
    float64 as numpy promotes them - against the oracle on the same bytes: block boundaries (byte positions of the
    file, tracking.py:107, 255) exactly, the series to rounding.  A channel that starts INSIDE a sample of the file
    reads the bytes of two samples, as the reference does (int32: every bit pattern is a finite value).  Cooperating
    workgroups and one workgroup per channel agree; a short record ends the run like the reference's short read."""
    m = pkg()
    ms = 30
    n = m.Settings().samplesPerCode
    rec8 = m.synth.generate(m.synth.Scene.default(), m.synth.record_length(n, ms))
    a8 = orc.acquire(orc.OracleSettings(), rec8[:11 * n])
    ch = orc.pre_run(orc.OracleSettings(numberOfChannels=3), a8)
    x = rec8.astype(np.float64)
    cases = [("float32", (x * 0.37 + 0.011).astype("<f4"), 0),
             ("float64", x * 1.234567890123e-3 - 7e-5, 0),
             ("float16", (x * 0.25).astype("<f2"), 0),
             ("uint16", (rec8.astype(np.int32) * 3 + 3000).astype("<u2"), 0),
             ("int32", rec8.astype("<i4") * 70001, 0),
             ("int32", rec8.astype("<i4") * 70001, 1),          # start bytes 1 (mod 4): straddling samples
             ("int64", rec8.astype("<i8") * 5000000001, 0),
             ("uint32", (rec8.astype(np.int64) * 1000 + 2 ** 31).astype("<u4"), 0)]
    for dtype, arr, shift in cases:
        isz = arr.dtype.itemsize
        skip = 16 * isz
        raw = np.concatenate([np.zeros(16, arr.dtype), arr])
        phase = (np.asarray(ch["codePhase"], dtype=np.int64) - 1) * isz + shift     # bytes behind skip
        s = m.Settings()
        s.dataType, s.numberOfChannels, s.msToProcess, s.skipNumberOfBytes = dtype, 3, float(ms), skip
        so = orc.OracleSettings(numberOfChannels=3, msToProcess=float(ms), dataType=dtype, skipNumberOfBytes=skip)
        want = orc.stack_series(orc.track(so, dict(PRN=ch["PRN"], acquiredFreq=ch["acquiredFreq"],
                                                   codePhase=phase.astype(np.float64), status=['T'] * 3), raw))
        a = m.AcquisitionResult(s, device=0)
        a._channels = np.rec.fromarrays([ch["PRN"], ch["acquiredFreq"], phase.astype(np.float64), ['T'] * 3],
                                        names='PRN,acquiredFreq,codePhase,status')
        path = str(tmp_path / ("rec." + dtype))
        raw.tofile(path)
        t = m.TrackingResult(a, device=0)
        with open(path, "rb") as fid:
            t.track(fid)
            assert fid.tell() == int(want[-1, 0, ms - 1])
        ctx = m.engine.get_context(s, 0)
        floaty = dtype in ("float32", "float64")
        assert ctx.timing()["track_kernel"] == (2 if floaty else 6), dtype
        scale = np.abs(want[:, 3:9]).max()
        assert np.array_equal(t.series[:, 0], want[:, 0]), (dtype, shift)
        assert _trk_err(t.series, want) < TRK_TOL, (dtype, shift, _trk_err(t.series, want), scale)
        if floaty:
            # the same record on the per-sample kernel, and with one workgroup per channel on the typed one
            for env, kern in (({"SGX_TRK_FLOAT_TYPED": "0"}, 6), ({"SGX_TRK_SPLIT": "1"}, 2)):
                os.environ.update(env)
                try:
                    tq = m.TrackingResult(a, device=0)
                    tq.track(m.DeviceFile(ctx.upload_bytes(raw.view(np.int8)), 0))
                finally:
                    for k in env:
                        os.environ.pop(k)
                assert ctx.timing()["track_kernel"] == kern, (dtype, env)
                assert np.array_equal(tq.series[:, 0], want[:, 0]) and _trk_err(tq.series, want) < TRK_TOL, (dtype, env)
        if dtype == "float32":
            os.environ["SGX_TRK_FLOAT_TYPED"] = "0"      # (the checks below: the per-sample kernel's layouts)
            os.environ["SGX_TRK_SPLIT"] = "1"
            try:
                t1 = m.TrackingResult(a, device=0)
                t1.track(m.DeviceFile(ctx.upload_bytes(raw.view(np.int8)), 0))
            finally:
                os.environ.pop("SGX_TRK_SPLIT")
                os.environ.pop("SGX_TRK_FLOAT_TYPED")
            assert ctx.timing()["track_members"] == 1 and ctx.timing()["track_kernel"] == 6
            assert np.array_equal(t1.series[:, 0], want[:, 0]) and _trk_err(t1.series, want) < TRK_TOL
            # the record ends inside the run: nothing is set, like the reference's short read (tracking.py:159-163)
            raw[:16 + 20 * n].tofile(path)
            t2 = m.TrackingResult(a, device=0)
            with open(path, "rb") as fid:
                assert t2.track(fid) is None and t2.series is None


def test_track_float32_record_with_an_outlier_takes_the_per_sample_kernel():
    """The typed kernel's fixed point is cut for samples of comparable size; a float record whose largest sample towers
    2^12 times above the mean |x| goes to the per-sample kernel (plain float64 sums, like the reference's) - and still
    agrees with the oracle."""
    m = pkg()
    ms = 20
    s = m.Settings()
    s.dataType, s.numberOfChannels, s.msToProcess = 'float32', 3, float(ms)
    n = s.samplesPerCode
    rec8 = m.synth.generate(m.synth.Scene.default(), m.synth.record_length(n, ms))
    a8 = orc.acquire(orc.OracleSettings(), rec8[:11 * n])
    ch = orc.pre_run(orc.OracleSettings(numberOfChannels=3), a8)
    arr = (rec8.astype(np.float64) * 0.37 + 0.011).astype("<f4")
    arr[5 * n + 123] = np.float32(3.0e6)
    phase = (np.asarray(ch["codePhase"], dtype=np.int64) - 1) * 4
    so = orc.OracleSettings(numberOfChannels=3, msToProcess=float(ms), dataType='float32')
    want = orc.stack_series(orc.track(so, dict(PRN=ch["PRN"], acquiredFreq=ch["acquiredFreq"],
                                               codePhase=phase.astype(np.float64), status=['T'] * 3), arr))
    ctx = m.engine.get_context(s, 0)
    chans = [(int(ch["PRN"][i]), float(ch["acquiredFreq"][i]), float(phase[i])) for i in range(3)]
    got, done = ctx.track(ctx.upload_bytes(arr.view(np.int8)), chans, ms, data_type=m._native.DT_FLOAT32)
    assert ctx.timing()["track_kernel"] == 6 and np.all(done == ms)
    assert np.array_equal(got[:, 0], want[:, 0]) and _trk_err(got, want) < TRK_TOL


def test_track_float32_where_a_group_meets_two_switches():
    """26 Msps: 12.7 samples per half chip, so a 16-sample group often holds the prompt ramp's switch AND the early / late
    one - the per-sample kernel's group path then takes the samples one by one with the switches as compares; 61.38 Msps:
    the group sums.  Arbitrary float32 values against the oracle."""
    m = pkg()
    for fs, IF in ((26000000.0, 6500000.0), (61380000.0, 15345000.0)):
        ms = 30
        s = m.Settings()
        os_ = orc.OracleSettings()
        for o in (s, os_):
            o.samplingFreq, o.IF, o.numberOfChannels, o.msToProcess, o.dataType = fs, IF, 2, float(ms), 'float32'
        n = s.samplesPerCode
        sc = m.synth.Scene.make(0xFE200 + n, fs, IF, [2, 5], [1750.0, -3300.0], [n // 3, n - 5], [9, 8])
        rec8 = m.synth.generate(sc, m.synth.record_length(n, ms))
        arr = (rec8.astype(np.float64) * 0.173 - 0.02).astype("<f4")
        prn = np.array([2, 5])
        freq = np.array([IF + 1750.0, IF - 3300.0])
        phase = (np.array([n // 3, n - 5], dtype=np.int64)) * 4
        want = orc.stack_series(orc.track(os_, dict(PRN=prn, acquiredFreq=freq, codePhase=phase.astype(np.float64),
                                                    status=['T'] * 2), arr))
        ctx = m.engine.get_context(s, 0)
        chans = [(int(prn[i]), float(freq[i]), float(phase[i])) for i in range(2)]
        for env, kern in (({"SGX_TRK_FLOAT_TYPED": "0"}, 6), ({}, 2)):
            os.environ.update(env)
            try:
                got, done = ctx.track(ctx.upload_bytes(arr.view(np.int8)), chans, ms, data_type=m._native.DT_FLOAT32)
            finally:
                for k in env:
                    os.environ.pop(k)
            assert ctx.timing()["track_kernel"] == kern and np.all(done == ms)
            assert np.array_equal(got[:, 0], want[:, 0]) and _trk_err(got, want) < TRK_TOL, (fs, kern)


def test_track_low_rate_int16_and_uint8_against_the_oracle():
    """int16 and uint8 records at a sampling rate below 16 x the chip rate (5.456 MHz: 5.3 samples per chip) - which the
    typed kernels exclude and round 3 refused - on the per-sample kernel, against the oracle."""
    m = pkg()
    fs, IF, ms = 5456000.0, 1364000.0, 40
    s = m.Settings()
    os_ = orc.OracleSettings()
    for o in (s, os_):
        o.samplingFreq, o.IF, o.numberOfChannels, o.msToProcess = fs, IF, 2, float(ms)
    n = s.samplesPerCode
    sc = m.synth.Scene.make(0xFE100 + n, fs, IF, [2, 5], [1750.0, -3300.0], [n // 3, n - 5], [9, 8])
    rec8 = m.synth.generate(sc, m.synth.record_length(n, ms))
    os_.acqSatelliteList = range(1, 7)
    a8 = orc.acquire(os_, rec8[:11 * n])
    ch = orc.pre_run(os_, a8)
    nch = len(ch["PRN"])
    for dtype, arr in (("int16", rec8.astype("<i2") * 129 - 5), ("uint8", (rec8.astype(np.int16) + 128).astype(np.uint8))):
        isz = arr.dtype.itemsize
        phase = (np.asarray(ch["codePhase"], dtype=np.int64) - 1) * isz
        for o in (s, os_):
            o.dataType, o.skipNumberOfBytes = dtype, 0
        want = orc.stack_series(orc.track(os_, dict(PRN=ch["PRN"], acquiredFreq=ch["acquiredFreq"],
                                                    codePhase=phase.astype(np.float64), status=['T'] * nch), arr))
        ctx = m.engine.get_context(s, 0)
        chans = [(int(ch["PRN"][i]), float(ch["acquiredFreq"][i]), float(phase[i])) for i in range(nch)]
        code = {"int16": m._native.DT_INT16, "uint8": m._native.DT_UINT8}[dtype]
        got, done = ctx.track(ctx.upload_bytes(arr.view(np.int8)), chans, ms, data_type=code)
        assert ctx.timing()["track_kernel"] == 6 and np.all(done == ms)
        assert np.array_equal(got[:, 0], want[:, 0]) and _trk_err(got, want) < TRK_TOL, dtype


@pytest.mark.parametrize("seed", list(range(31, 43)))
def test_random_front_ends_and_scenes_against_oracle(seed):
    """Random sampling rate (other FFT factorisations: 26 000 = 2^4 5^3 13, 20 460 = 2^2 3 5 11 31, 12 276 =
    2^2 3^2 11 31), IF, Doppler up to the band edge, code phase, amplitude and loop settings."""
    m = pkg()
    rng = np.random.default_rng(seed)
    fs = float(rng.choice([38192000.0, 16367600.0, 26000000.0, 20460000.0, 12276000.0, 5456000.0]))
    IF = float(rng.choice([0.25, 0.2, 0.31]) * fs)
    s = m.Settings()
    os_ = orc.OracleSettings()
    kw = dict(samplingFreq=fs, IF=IF, acqSatelliteList=range(1, 6), numberOfChannels=2, msToProcess=30.0,
              dllCorrelatorSpacing=float(rng.choice([0.5, 0.5, 0.3, 0.7])), dllNoiseBandwidth=float(rng.choice([2.0, 1.0, 5.0])),
              pllNoiseBandwidth=float(rng.choice([25.0, 10.0, 50.0])), acqSearchBand=float(rng.choice([14.0, 14.0, 8.0])))
    for k, v in kw.items():
        setattr(s, k, v)
        setattr(os_, k, v)
    n = s.samplesPerCode
    half = kw["acqSearchBand"] * 500.0
    prns = sorted(rng.choice(np.arange(1, 6), size=2, replace=False).tolist())
    sc = m.synth.Scene.make(0xAB000 + seed, fs, IF, prns, [float(rng.uniform(-half, half)) for _ in prns],
                            [int(rng.integers(0, n)) for _ in prns], [int(rng.integers(4, 10)) for _ in prns])
    rec = m.synth.generate(sc, m.synth.record_length(n, 30))
    _oracle_vs_gpu(m, s, os_, rec, 30)


def test_tracking_channels_without_a_signal(default_record):
    """Channels started on PRNs that are not in the record (noise only: the loops wander, the discriminators see
    arbitrary ratios) and with a 4 kHz frequency error: the GPU must still follow the reference's arithmetic."""
    m = pkg()
    g = load_golden("trk_default.npz")
    s = m.Settings()
    s.numberOfChannels = 3
    s.msToProcess = 150.0
    freqs = np.array([9548000.0 + 321.0, 9548000.0 - 2750.0, float(g["ch_acquiredFreq"][0]) + 4000.0])
    prn = np.array([9, 28, int(g["ch_PRN"][0])])
    phase = np.array([17.0, 30011.0, float(g["ch_codePhase"][0])])
    a = m.AcquisitionResult(s, device=0)
    a._channels = np.rec.fromarrays([prn, freqs, phase, ['T'] * 3], names='PRN,acquiredFreq,codePhase,status')
    t = m.TrackingResult(a, device=0)
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload(default_record[:160 * 38193])
    t.track(m.DeviceFile(rec))
    rec.free()
    os_ = orc.OracleSettings(numberOfChannels=3, msToProcess=150.0)
    want = orc.stack_series(orc.track(os_, dict(PRN=prn, acquiredFreq=freqs, codePhase=phase, status=np.array(['T'] * 3)),
                                      default_record[:160 * 38193]))
    assert np.array_equal(t.series[:, 0], want[:, 0])
    # without a signal the sums are noise of magnitude ~1e4: compare on that scale
    scale = np.sqrt(np.mean(want[:, 3:9] ** 2, axis=(1, 2)))
    err = np.max(np.abs(t.series[:, 3:9] - want[:, 3:9]), axis=(1, 2)) / scale
    assert err.max() < 1e-6, err
    assert np.max(np.abs(t.series[:, 1:3] - want[:, 1:3])) < 1e-5


def test_rccl_communicator_single_rank_roundtrip():
    """The acquisition peak gather's native transport (sgx_comm_*: dlopen of librccl, ncclCommInitRank,
    ncclAllGather on the context stream) with one rank on device 0: what it gathers is what was sent."""
    m, s, ctx = _ctx()
    sh = pkg("shard")
    comm = m._native.Comm(ctx, 1, 0, m._native.Comm.unique_id())
    try:
        g = load_golden("acq_default.npz")
        mine = list(range(32))
        res = dict(carrFreq=g["carrFreq"], codePhase=g["codePhase"], peakMetric=g["peakMetric"],
                   freqBin=g["freqBin"], fineIdx=g["fineIdx"])
        buf = sh.pack_peaks(mine, res, 32)
        gather = sh.RcclGather(comm)
        for _ in range(3):
            out = gather.allgather(buf)
            assert out.shape == (1, 32)
            assert out.tobytes() == buf.tobytes()
        merged = sh.merge_peaks(out)
        assert np.array_equal(merged["codePhase"], g["codePhase"])
        # the sharded acquisition entry point over this transport equals the plain call
        a = m.AcquisitionResult(s, device=0)
        rec = ctx.synth(m.synth.Scene.default(), 11 * s.samplesPerCode)
        sig = m.DeviceSignal(rec, 0, 11 * s.samplesPerCode)
        sh.acquire_sharded(a, sig, 0, 1, gather)      # (sgx_acquire_sharded: packed on the device, ncclAllGather, one look)
        b = m.AcquisitionResult(s, device=0)
        b.acquire(sig)
        for f in ("codePhase", "carrFreq", "peakMetric"):
            assert np.array_equal(a.results[f], b.results[f]), f
        for f in ("freqBin", "fineIdx"):
            assert np.array_equal(a.internals[f], b.internals[f]), f
        # one rank's shard of a world of eight, run alone (no communicator): its own PRNs, the plain call's values
        for rk in (0, 3, 7):
            c = m.AcquisitionResult(s, device=0)
            sh.acquire_sharded(c, sig, rk, 8, sh.LocalGather())
            mine = list(sh.plan_shards(32, 8)[rk])
            other = [p for p in range(32) if p not in mine]
            for f in ("codePhase", "carrFreq", "peakMetric"):
                assert np.array_equal(c.results[f][mine], b.results[f][mine]), (rk, f)
                assert not np.any(c.results[f][other])
            assert np.array_equal(c.internals["fineIdx"][mine], b.internals["fineIdx"][mine])
        rec.free()
    finally:
        comm.close()


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N devices it must fail, not report n_gpus 1."""
    import subprocess
    import sys
    from conftest import ROOT
    n = pkg()._native.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode != 0
    assert b"n_gpus" not in r.stdout
    assert b"refusing" in r.stderr


def test_acquire_accepts_any_real_valued_signal(default_record):
    """acquisition.py:55-59 works on whatever real dtype it is handed.  A non-int8 longSignal goes to HBM as fp64
    (sgx_acquire_f64); indices bit-exact and the metric within 1e-9 of the oracle on the SAME array: a rescaled and
    offset float record, an int16-valued one, and the int8 record as float64 (must equal the int8 path exactly)."""
    m = pkg()
    s = m.Settings()
    n = 11 * s.samplesPerCode
    x8 = default_record[:n]
    ref8 = m.AcquisitionResult(s, device=0)
    ref8.acquire(x8)
    a = m.AcquisitionResult(s, device=0)
    a.acquire(x8.astype(np.float64))
    assert np.array_equal(a.codePhase, ref8.codePhase) and np.array_equal(a.carrFreq, ref8.carrFreq)
    assert np.array_equal(a.internals["freqBin"], ref8.internals["freqBin"])
    assert np.allclose(a.peakMetric, ref8.peakMetric, rtol=1e-12, atol=0)
    for sig in (x8.astype(np.float64) * 0.37 + 0.123, x8.astype(np.int16) * 211, (x8.astype(np.float32) - 0.5)):
        os_ = orc.OracleSettings(acqSatelliteList=list(range(1, 13)))
        s12 = m.Settings()
        s12.acqSatelliteList = list(range(1, 13))
        want = orc.acquire(os_, np.asarray(sig, dtype=np.float64))
        b = m.AcquisitionResult(s12, device=0)
        b.acquire(sig)
        assert np.array_equal(b.codePhase[:12], want["codePhase"][:12])
        assert np.array_equal(b.carrFreq[:12], want["carrFreq"][:12])
        assert np.array_equal(b.internals["freqBin"][:12], want["freqBin"][:12])
        assert np.allclose(b.peakMetric[:12], want["peakMetric"][:12], rtol=1e-9, atol=0)
    with pytest.raises(TypeError):
        m.AcquisitionResult(s, device=0).acquire(x8.astype(np.complex128))


def _step(m, s, rec, ms, deferred):
    n = s.samplesPerCode
    a = m.AcquisitionResult(s, device=0, deferred=deferred)
    a.acquire(m.DeviceSignal(rec, 0, 11 * n))
    a.preRun()
    t = m.TrackingResult(a, device=0)
    fid = m.DeviceFile(rec)
    t.track(fid)
    return a, t, fid


def test_deferred_step_equals_the_eager_one(default_record):
    """acquire -> preRun -> track queued without a look in between (sgx_acquire_begin / sgx_track_chained: preRun on the
    device, one wait) against the eager calls: every result bit for bit, 8 channels and a table with channels off."""
    m = pkg()
    # 8 channels, all on; 5 channels for 8 detections; 8 channels for the 4 detections among PRN indices 0..10 (channels
    # off); 12 channels (more than 8: the queued sequence does not apply, the same objects run the eager calls)
    for nch, ms, n_search, chained in ((8, 300, 32, True), (5, 150, 32, True), (8, 150, 11, True), (12, 120, 32, False)):
        s = m.Settings()
        s.numberOfChannels = nch
        s.msToProcess = float(ms)
        s.acqSatelliteList = list(range(1, n_search + 1))
        ctx = m.engine.get_context(s, 0)
        rec = ctx.upload(default_record)
        ae, te, fe = _step(m, s, rec, ms, False)
        ad, td, fd = _step(m, s, rec, ms, True)
        assert td.chained == chained and not te.chained
        assert (ad._pending is not None) == chained   # chained: tracking is done and nobody has looked at the search yet
        assert np.array_equal(td.series, te.series)
        assert fd.tell() == fe.tell()
        for f in ("carrFreq", "codePhase", "peakMetric"):
            assert np.array_equal(ad.results[f], ae.results[f]), f
        assert ad._pending is None
        for f in ("freqBin", "fineIdx"):
            assert np.array_equal(ad.internals[f], ae.internals[f]), f
        for f in ("PRN", "acquiredFreq", "codePhase", "status"):
            assert np.array_equal(ad.channels[f], ae.channels[f]), f
        assert ad.channels.PRN.dtype == ae.channels.PRN.dtype
        assert len(td.results) == len(te.results) == int(np.sum(ae.channels.PRN != 0))
        for name in td.results.dtype.names:
            for j in range(len(te.results)):
                assert np.array_equal(td.results[j][name], te.results[j][name]), name
        rec.free()


def test_deferred_results_looked_at_early_and_superseded(default_record):
    """A deferred search somebody looks at before tracking simply becomes an eager one; one search is pending per context."""
    m = pkg()
    s = m.Settings()
    s.msToProcess = 60.0
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload(default_record)
    n = s.samplesPerCode
    want = m.AcquisitionResult(s, device=0)
    want.acquire(m.DeviceSignal(rec, 0, 11 * n))
    want.preRun()
    a = m.AcquisitionResult(s, device=0, deferred=True)
    a.acquire(m.DeviceSignal(rec, 0, 11 * n))
    assert np.array_equal(a.peakMetric, want.peakMetric) and a._pending is None     # the look
    a.preRun()
    assert np.array_equal(a.channels.PRN, want.channels.PRN)
    t = m.TrackingResult(a, device=0)
    t.track(m.DeviceFile(rec))
    assert not t.chained and len(t.results) == 8
    # preRun() asked for while queued, channels looked at before tracking: the host's preRun
    b = m.AcquisitionResult(s, device=0, deferred=True)
    b.acquire(m.DeviceSignal(rec, 0, 11 * n))
    b.preRun()
    assert b._prerun_pending
    assert np.array_equal(b.channels.acquiredFreq, want.channels.acquiredFreq) and not b._prerun_pending
    # two deferred searches on one context: the first one's owner is told
    c1 = m.AcquisitionResult(s, device=0, deferred=True)
    c1.acquire(m.DeviceSignal(rec, 0, 11 * n))
    c2 = m.AcquisitionResult(s, device=0, deferred=True)
    c2.acquire(m.DeviceSignal(rec, 0, 11 * n))
    with pytest.raises(RuntimeError):
        c1.results
    assert np.array_equal(c2.carrFreq, want.carrFreq)
    rec.free()


def test_deferred_search_raises_the_references_index_error_at_the_first_look():
    g = load_golden("acq_edges.npz")
    m = pkg()
    s = m.Settings()
    s.acqSatelliteList = [1]
    s.msToProcess = 20.0
    ctx = m.engine.get_context(s, 0)
    i = [k for k in range(len(g["phases"])) if str(g["err"][k]) == "IndexError"][0]
    x = m.synth.generate(scene_from_json(g["scenes"][i]), 40 * s.samplesPerCode)
    rec = ctx.upload(x)
    a = m.AcquisitionResult(s, device=0, deferred=True)
    a.acquire(m.DeviceSignal(rec, 0, 11 * s.samplesPerCode))      # queued: nothing raised yet
    a.preRun()
    t = m.TrackingResult(a, device=0)
    with pytest.raises(IndexError):
        t.track(m.DeviceFile(rec))                                 # the device found no channel table to make: the look raises
    rec.free()


def _clean_record(m, s, amp, n_ms, prn=5, doppler=1250.0, start=7000):
    """A NOISELESS one-satellite record: round(amp * chip * cos(carrier)) - every sample lines up with the replica."""
    n = s.samplesPerCode
    N = (n_ms + 2) * (n + 2)
    t = np.arange(N, dtype=np.float64)
    code = np.asarray(s.generateCAcode(prn - 1))
    chip = code[(np.floor((t - start) * (s.codeFreqBasis / s.samplingFreq)).astype(np.int64)) % 1023]
    x = np.rint(amp * chip * np.cos(2 * np.pi * ((s.IF + doppler) / s.samplingFreq) * t + 0.3))
    return x.astype(np.int64)


def test_a_record_too_strong_for_the_speculative_kernel_is_tracked_by_the_round_3_kernel(capfd):
    """csrc/sgx_trk3.hip exchanges unit sums as 48-bit payloads at 2^30: a unit's total must stay below 2^17.  A noiseless
    int8 record of amplitude 120 (mean magnitude 76) would pass that - and round 5's guard, which looked at payloads that had
    already wrapped.  Round 6 bounds the sums by the samples' magnitudes (one scan per resident record, csrc/sgx_trk.hip:
    if_mag_bound): the round-3 kernel tracks the record, SAID on stderr, and the results are the reference's.  Amplitude 70
    (unit sums of 72 000: past HALF the room) is sent there by the kernel's own look at its prompt sums, with a repeated
    launch that is said as well; amplitude 50 and an offset-binary uint8 record of amplitude 120 (whose scale is half) stay
    on the speculative kernel - and are the reference's too."""
    m = pkg()
    for amp, kind, kernel in ((120, "int8", 2), (70, "int8", 2), (50, "int8", 5), (120, "uint8", 5)):
        s = m.Settings()
        os_ = orc.OracleSettings()
        kw = dict(acqSatelliteList=range(1, 9), numberOfChannels=1, msToProcess=40.0)
        for k, v in kw.items():
            setattr(s, k, v)
            setattr(os_, k, v)
        x = _clean_record(m, s, amp, 40)
        ctx = m.engine.get_context(s, 0)
        if kind == "uint8":
            s.dataType = os_.dataType = "uint8"
            raw = (x + 128).astype(np.uint8)
            ref = orc.acquire(os_, raw[:11 * s.samplesPerCode])
            chans = orc.pre_run(os_, ref)
            a = m.AcquisitionResult(s, device=0)
            a.acquire(raw[:11 * s.samplesPerCode].astype(np.float64))     # (acquisition takes any real-valued signal)
            assert np.array_equal(a.carrFreq, ref["carrFreq"]) and np.array_equal(a.codePhase, ref["codePhase"])
            a.preRun()
            t = m.TrackingResult(a, device=0)
            rec = ctx.upload_bytes(raw)
            t.track(m.DeviceFile(rec))
            rec.free()
            want = orc.stack_series(orc.track(os_, chans, raw))
            assert np.array_equal(t.series[:, 0], want[:, 0])
            assert _trk_err(t.series, want) < TRK_TOL
        else:
            _oracle_vs_gpu(m, s, os_, x.astype(np.int8), 40)
        err = capfd.readouterr().err
        tm = ctx.timing()
        assert int(tm["track_kernel"]) == kernel, (amp, kind, tm)
        assert ("too strong for the speculative kernel" in err) == (kernel == 2), err


def test_deferred_step_on_a_short_record_leaves_the_results_unset(default_record):
    """The reference's short-read exit (tracking.py:159-163: message, fid.close(), results NOT set) through the queued
    sequence: the chained launch reports the blocks it completed, the eager one the same."""
    gs = load_golden("trk_short.npz")
    m = pkg()
    s = m.Settings()
    s.numberOfChannels = 4
    s.msToProcess = 400.0
    ctx = m.engine.get_context(s, 0)
    rec = ctx.upload(default_record[:int(gs["n_samples"])])
    outs = []
    for deferred in (False, True):
        a = m.AcquisitionResult(s, device=0, deferred=deferred)
        a.acquire(m.DeviceSignal(rec, 0, 11 * s.samplesPerCode))
        a.preRun()
        t = m.TrackingResult(a, device=0)
        fid = m.DeviceFile(rec)
        ret = t.track(fid)
        assert ret is None and fid.closed and not t.has_results() and t.chained == deferred
        with pytest.raises(AssertionError):
            t.results
        outs.append(np.array(a.channels.PRN))
    assert np.array_equal(outs[0], outs[1])
    rec.free()


def test_sharded_search_reports_the_references_index_error_from_the_rank_that_met_it():
    """sgx_acquire_sharded marks the record of a PRN whose coarse code phase equals the samples per chip (the reference's
    IndexError, acquisition.py:152-162): the rank whose share holds it raises - and, through the gather, so would every other
    rank; a rank whose share does not hold it, run alone, finishes."""
    g = load_golden("acq_edges.npz")
    m = pkg()
    sh = pkg("shard")
    s = m.Settings()
    ctx = m.engine.get_context(s, 0)
    i = [k for k in range(len(g["phases"])) if str(g["err"][k]) == "IndexError"][0]
    x = m.synth.generate(scene_from_json(g["scenes"][i]), 11 * s.samplesPerCode)     # PRN 1 at code phase 37
    rec = ctx.upload(x)
    sig = m.DeviceSignal(rec, 0, 11 * s.samplesPerCode)
    a = m.AcquisitionResult(s, device=0)
    with pytest.raises(IndexError):
        sh.acquire_sharded(a, sig, 0, 8, sh.LocalGather())        # PRN indices 0..3: the failing one is here
    b = m.AcquisitionResult(s, device=0)
    sh.acquire_sharded(b, sig, 3, 8, sh.LocalGather())            # PRN indices 12..15
    assert not np.any(b.carrFreq)
    comm = m._native.Comm(ctx, 1, 0, m._native.Comm.unique_id())
    try:
        c = m.AcquisitionResult(s, device=0)
        with pytest.raises(IndexError):
            sh.acquire_sharded(c, sig, 0, 1, sh.RcclGather(comm))  # ... and through a real (one-rank) gather
    finally:
        comm.close()
    rec.free()
