"""Pin the CPU oracle (oracle/softgnss_oracle.py) bit-for-bit against outputs of the reference
itself (tests/golden/*.npz, captured by tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, pkg, scene_from_json
from oracle import softgnss_oracle as orc


def test_ca_codes_match_reference_and_icd():
    g = load_golden("codes.npz")
    codes = np.stack([orc.generate_ca_code(p) for p in range(32)])
    assert np.array_equal(codes.astype(np.int8), g["ca_codes"])
    # IS-GPS-200 first-10-chips octal known answers (chip +1 <-> bit 1), SURVEY.md section 4
    octal = {1: 0o1440, 2: 0o1620, 3: 0o1710, 4: 0o1744, 19: 0o1633, 32: 0o1712}
    for prn, want in octal.items():
        bits = (codes[prn - 1][:10] > 0).astype(int)
        assert int("".join(map(str, bits)), 2) == want
    assert np.all(codes.sum(axis=1) == 1.0)
    with pytest.raises(AssertionError):
        orc.generate_ca_code(32)


def test_ca_table_and_constants():
    g = load_golden("codes.npz")
    s = orc.OracleSettings()
    assert s.samplesPerCode == int(g["samples_per_code"]) == 38192
    t = orc.make_ca_table(s)
    assert np.array_equal(np.packbits(t > 0, axis=1), g["ca_table_bits"])
    assert hashlib.sha256(t.astype(np.int8).tobytes()).hexdigest()[:16] == "9c8df822b4154971"
    assert orc.calc_loop_coef(2.0, 0.7, 1.0) == tuple(g["loop_dll"]) == (0.06984693877551021, 0.37)
    assert orc.calc_loop_coef(25.0, 0.7, 0.25) == tuple(g["loop_pll"])
    assert len(orc.freq_bins(s)) == 29


def test_acquire_prn1_matches_reference(default_record):
    g = load_golden("acq_prn1.npz")
    s = orc.OracleSettings(acqSatelliteList=[1])
    r = orc.acquire(s, default_record[:int(g["n_samples"])])
    for k in ("carrFreq", "codePhase", "peakMetric"):
        assert np.array_equal(r[k], g[k]), k
    assert r["freqBin"][0] == g["freqBin"][0] and r["fineIdx"][0] == g["fineIdx"][0]


def test_acquire_as_written_equals_hoisted(default_record):
    s = orc.OracleSettings(acqSatelliteList=[1])
    a = orc.acquire(s, default_record[:11 * 38192], as_written=True)
    b = orc.acquire(s, default_record[:11 * 38192])
    for k in ("carrFreq", "codePhase", "peakMetric"):
        assert np.array_equal(a[k], b[k])


@pytest.mark.slow
def test_acquire_all_prns_and_prerun_match_reference(default_record):
    g = load_golden("acq_default.npz")
    s = orc.OracleSettings()
    r = orc.acquire(s, default_record[:int(g["n_samples"])])
    for k in ("carrFreq", "codePhase", "peakMetric", "freqBin"):
        assert np.array_equal(r[k], g[k]), k
    det = g["carrFreq"] > 0
    assert np.array_equal(r["fineIdx"][det], g["fineIdx"][det])
    ch = orc.pre_run(s, r)
    assert np.array_equal(ch["PRN"], g["ch_PRN"])
    assert np.array_equal(ch["acquiredFreq"], g["ch_acquiredFreq"])
    assert np.array_equal(ch["codePhase"], g["ch_codePhase"])
    assert list(ch["status"]) == [str(x) for x in g["ch_status"]]


def test_acquire_code_phase_edges():
    g = load_golden("acq_edges.npz")
    synth = pkg("synth")
    s = orc.OracleSettings(acqSatelliteList=[1])
    for i, c in enumerate(g["phases"]):
        if c not in (0, 37, 38191):       # keep the CPU suite short: first/last/IndexError cases
            continue
        x = synth.generate(scene_from_json(g["scenes"][i]), 11 * s.samplesPerCode)
        if str(g["err"][i]) == "IndexError":
            with pytest.raises(IndexError):
                orc.acquire(s, x)
            continue
        r = orc.acquire(s, x)
        assert r["codePhase"][0] == g["codePhase"][i] == c
        assert r["carrFreq"][0] == g["carrFreq"][i]
        assert r["peakMetric"][0] == g["peakMetric"][i]
        assert r["freqBin"][0] == g["freqBin"][i] and r["fineIdx"][0] == g["fineIdx"][i]


def test_track_matches_reference(default_record):
    g = load_golden("trk_default.npz")
    ms = int(g["ms"])
    s = orc.OracleSettings(numberOfChannels=4, msToProcess=float(ms))
    ch = dict(PRN=g["ch_PRN"], acquiredFreq=g["ch_acquiredFreq"], codePhase=g["ch_codePhase"],
              status=['T'] * 4)
    out = orc.track(s, ch, default_record)
    assert out is not None and len(out) == 4
    got = orc.stack_series(out)
    assert got.shape == g["series"].shape
    assert np.array_equal(got, g["series"])          # bit-for-bit: same numpy, same op order
    assert [o["PRN"] for o in out] == list(g["PRN"])
    # first-block remCodePhase known answer shows up as block 2 size
    assert got[0, 0, 0] == g["ch_codePhase"][0] + 38192


def test_track_int16_matches_reference():
    """Settings.dataType = 'int16' (tracking.py:154): samples of two bytes, byte seeks and byte positions as the
    reference has them (tracking.py:107,255).  A locked channel and two channels started where the byte seek lands."""
    g = load_golden("trk_int16.npz")
    synth = pkg("synth")
    rec16 = (synth.generate(scene_from_json(g["scene"]), int(g["n_samples"])).astype(np.int16) * int(g["scale"])).astype("<i2")
    acq = orc.acquire(orc.OracleSettings(), rec16[:11 * 38192])
    assert np.array_equal(acq["codePhase"], g["codePhase"]) and np.array_equal(acq["carrFreq"], g["carrFreq"])
    for case in ("locked", "as_is"):
        nch = len(g[case + "_PRN"])
        s = orc.OracleSettings(numberOfChannels=nch, msToProcess=float(g["ms"]), dataType='int16',
                               skipNumberOfBytes=int(g[case + "_skip"]))
        ch = dict(PRN=g[case + "_PRN"], acquiredFreq=g[case + "_acquiredFreq"], codePhase=g[case + "_codePhase"],
                  status=['T'] * nch)
        out = orc.track(s, ch, rec16)
        assert out is not None
        assert np.array_equal(orc.stack_series(out), g[case + "_series"]), case


def test_track_float32_matches_reference():
    """Settings.dataType = 'float32' (tracking.py:154): a record of floats (200 x + 7) / 32768 in a file with a
    4000-byte header, channels on whole samples; byte seeks and byte positions as the reference has them."""
    g = load_golden("trk_float32.npz")
    synth = pkg("synth")
    rec8 = synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]))
    recf = ((rec8.astype(np.int32) * 200 + 7) / 32768.0).astype("<f4")
    raw = np.concatenate([np.zeros(int(g["skip"]) // 4, "<f4"), recf])
    nch = len(g["PRN"])
    s = orc.OracleSettings(numberOfChannels=nch, msToProcess=float(g["ms"]), dataType='float32', skipNumberOfBytes=int(g["skip"]))
    ch = dict(PRN=g["PRN"], acquiredFreq=g["acquiredFreq"], codePhase=g["codePhase"], status=['T'] * nch)
    out = orc.track(s, ch, raw)
    assert out is not None
    assert np.array_equal(orc.stack_series(out), g["series"])


def test_track_short_read_returns_none(default_record):
    g = load_golden("trk_short.npz")
    gt = load_golden("trk_default.npz")
    assert bool(g["returned_none"]) and bool(g["results_unset"]) and bool(g["closed"])
    s = orc.OracleSettings(numberOfChannels=4, msToProcess=400.0)
    ch = dict(PRN=gt["ch_PRN"], acquiredFreq=gt["ch_acquiredFreq"], codePhase=gt["ch_codePhase"],
              status=['T'] * 4)
    assert orc.track(s, ch, default_record[:int(g["n_samples"])]) is None


def test_second_front_end_matches_reference():
    """16.3676 Msps / IF 4.1304 MHz (samplesPerCode 16368): table, acquisition of 12 PRNs, preRun and
    3 channels x 250 ms of tracking, all bit-for-bit against the reference's outputs."""
    g = load_golden("rate2.npz")
    synth = pkg("synth")
    s = orc.OracleSettings(samplingFreq=16367600.0, IF=4130400.0, msToProcess=250.0, numberOfChannels=3,
                           acqSatelliteList=range(1, 13))
    n = s.samplesPerCode
    assert n == int(g["samples_per_code"]) == 16368
    assert np.array_equal(np.packbits(orc.make_ca_table(s) > 0, axis=1), g["ca_table_bits"])
    rec = synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]))
    r = orc.acquire(s, rec[:11 * n])
    for k in ("carrFreq", "codePhase", "peakMetric", "freqBin"):
        assert np.array_equal(r[k], g[k]), k
    det = g["carrFreq"] > 0
    assert list(np.flatnonzero(det) + 1) == [2, 5, 9]
    assert np.array_equal(r["fineIdx"][det], g["fineIdx"][det])
    ch = orc.pre_run(s, r)
    assert np.array_equal(ch["PRN"], g["ch_PRN"]) and np.array_equal(ch["codePhase"], g["ch_codePhase"])
    out = orc.track(s, ch, rec)
    assert np.array_equal(orc.stack_series(out), g["series"])


def test_find_preambles_matches_reference():
    """postNavigation.findPreambles / navPartyChk restated; golden = the reference's own result on the I_P
    series of its own tracker (structured navigation data, subframes every 6000 ms)."""
    g = load_golden("nav_preambles.npz")
    first, active = orc.find_preambles(g["I_P"], ['T', 'T'], 2)
    assert np.array_equal(first, g["firstSubFrame"]) and list(first) == [1999, 1999]
    assert np.array_equal(active, g["activeChnList"])
    # a channel without preambles (scrambled signs) drops out of the active list
    rng = np.random.default_rng(3)
    noise = g["I_P"][1] * rng.choice([-1.0, 1.0], size=g["I_P"].shape[1])
    first2, active2 = orc.find_preambles(np.stack([g["I_P"][0], noise]), ['T', 'T'], 2)
    assert first2[0] == 1999 and first2[1] == 0 and list(active2) == [0]


def test_nav_parity_check_on_generated_words():
    synth = pkg("synth")
    bits = synth.subframe_bits(77, first_boundary=0, n_bits=900).astype(np.float64) * 2 - 1
    for w in range(1, 29):
        ndat = bits[30 * w - 2:30 * w + 30].copy()
        st = orc.nav_party_chk(ndat)
        assert st in (1, -1) and st == -ndat[1]
        bad = bits[30 * w - 2:30 * w + 30].copy()
        bad[7] *= -1
        assert orc.nav_party_chk(bad) == 0


def test_probe_statistics_match_reference():
    """Settings.probeData's Welch PSD and histogram (initialize.py:330-417), captured from the reference itself."""
    g = load_golden("probe_default.npz")
    synth = pkg("synth")
    data = synth.generate(scene_from_json(g["scene"]), int(g["n_samples"]))
    f, pxx, hist = orc.probe_stats(orc.OracleSettings(), data)
    assert np.array_equal(f, g["f"]) and np.array_equal(pxx, g["Pxx"]) and np.array_equal(hist, g["hist"])
    assert hist.sum() == data.size and len(hist) == 255
    assert np.array_equal(data[1:38192 // 50], g["time_amp"])
