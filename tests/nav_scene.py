"""A physically consistent synthetic scene for the position-fix tests: orbits, a receiver position, signal travel
times (Earth rotation and the troposphere model included), Dopplers from the range rates, and the navigation
message carrying the very ephemerides the geometry was computed from.  Test infrastructure (uses the oracle)."""
import numpy as np

import oracle.softgnss_oracle as orc
from conftest import pkg

C_MPS = 299792458.0
L1 = 1575.42e6
BIT_CHIPS = 20 * 1023


def geodetic_to_ecef(lat_deg, lon_deg, h):
    a, f = 6378137.0, 1 / 298.257223563
    e2 = f * (2 - f)
    la, lo = np.radians(lat_deg), np.radians(lon_deg)
    n = a / np.sqrt(1 - e2 * np.sin(la) ** 2)
    return np.array([(n + h) * np.cos(la) * np.cos(lo), (n + h) * np.cos(la) * np.sin(lo), (n * (1 - e2) + h) * np.sin(la)])


def _arrival(t_sv, prn, table, rx):
    """GPS time at which the signal leaving satellite `prn` at satellite-clock time t_sv reaches rx, plus elevation."""
    pos, clk = orc.satpos(t_sv, [prn], table)
    x = pos[:, 0]
    tau = 0.07
    for _ in range(4):
        rot = orc.e_r_corr(tau, x)
        tau = np.linalg.norm(rot - rx) / C_MPS
    az, el, _ = orc.topocent(rx, rot - rx)
    trop = orc.tropo(np.sin(np.radians(el)), 0.0, 1013.0, 293.0, 50.0, 0.0, 0.0, 0.0)
    return t_sv - clk[0] + tau + trop / C_MPS, el


def _gdop(units):
    A = np.hstack([-np.asarray(units), np.ones((len(units), 1))])
    try:
        return float(np.sqrt(np.trace(np.linalg.inv(A.T.dot(A)))))
    except np.linalg.LinAlgError:
        return np.inf


def build(seed=2024, fs=16368000.0, IF=4130400.0, site=(40.0, -105.2, 1650.0), tow0=16800, first_boundary=260,
          n_sats=6, el_mask=12.0, amp=8):
    """-> (scene, truth).  fs is a whole number of samples per millisecond on purpose: the reference turns sample
    counts into milliseconds by dividing by samplesPerCode (postNavigation.py:62), which is exact only then.  The
    n_sats satellites with the smallest GDOP among those above el_mask are used.  The subframe that starts at table bit `first_boundary` leaves every satellite at
    satellite time 6 * tow0; the earliest arrival lands first_boundary * 20 ms (+ 100 samples) into the record."""
    synth = pkg("synth")
    rx = geodetic_to_ecef(*site)
    t_sv = 6.0 * tow0
    table = np.zeros((32, 27))
    ephs = {}
    for prn in range(1, 33):
        e = synth.make_ephemeris(seed * 100 + prn, toe=6 * tow0)
        bits = synth.nav_message_bits(1, 0, 2048, tow0, 1, e)
        dec, _ = orc.ephemeris([str(int(b)) for b in bits[300:1800]], str(int(bits[299])))
        table[prn - 1] = [float(v) for v in dec]
        ephs[prn] = e
    import itertools
    cand = []
    for prn in range(1, 33):
        t_arr, el = _arrival(t_sv, prn, table, rx)
        if el > el_mask:
            pos, _ = orc.satpos(t_sv, [prn], table)
            u = pos[:, 0] - rx
            cand.append((prn, t_arr, el, u / np.linalg.norm(u)))
    assert len(cand) >= n_sats, "not enough satellites in view for this seed"
    cand = cand[:14]
    best = min(itertools.combinations(range(len(cand)), n_sats), key=lambda c: _gdop([cand[i][3] for i in c]))
    gdop = _gdop([cand[i][3] for i in best])
    cand = [cand[i][:3] for i in best]
    prns = [c[0] for c in cand]
    t0 = min(c[1] for c in cand) - (first_boundary * 0.020 + 100.0 / fs)      # GPS time of sample 0
    dop, arrivals = [], []
    for prn, t_arr, el in cand:
        t_arr2, _ = _arrival(t_sv + 1.0, prn, table, rx)
        range_rate = (t_arr2 - t_arr - 1.0) * C_MPS
        dop.append(-range_rate * L1 / C_MPS)
        arrivals.append((t_arr - t0) * fs)
    sc = synth.Scene.make(seed, fs, IF, prns, dop, [0] * n_sats, [amp] * n_sats)
    period = synth.NAV_TABLE_BITS * BIT_CHIPS << 32
    for s, n_s in zip(sc.sats, arrivals):
        c0 = (first_boundary * BIT_CHIPS << 32) - int(round(n_s * s["code_fcw"]))
        s["code_c0"] = c0 % period
    sc = sc.with_nav_message(first_boundary, tow0, 1, {p: ephs[p] for p in prns})
    truth = dict(rx=rx, site=site, prns=prns, doppler=dop, arrival_samples=arrivals, eph_table=table,
                 tow=6 * tow0, first_boundary=first_boundary, elevation=[c[2] for c in cand], gdop=gdop)
    return sc, truth
