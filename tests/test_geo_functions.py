"""Satellite positions, least-squares fix and coordinate conversions: the oracle and the
C-ABI host code against outputs of the reference's geoFunctions (tests/golden/geo_cases.npz, made by
tests/golden/make_golden.py on transmitted-and-decoded ephemerides).  No GPU involved: this is scalar host code.

Bars: satellite positions 1e-6 m, clocks 1e-15 s, receiver position 1e-6 m, angles 1e-10 deg, DOP 1e-10,
geodetic 1e-11 deg / 1e-6 m, UTM 1e-6 m (libm vs numpy transcendental functions differ by an ulp)."""
import numpy as np
import pytest

import oracle.softgnss_oracle as orc
from conftest import load_golden, pkg


class _S(object):
    c = 299792458.0
    useTropCorr = True


def _impls():
    gf = pkg("geoFunctions")

    class Product(object):
        satpos = staticmethod(lambda t, prn, tab: gf.satpos(t, prn, tab, _S()))
        topocent = staticmethod(gf.topocent)
        togeod = staticmethod(gf.togeod)
        tropo = staticmethod(gf.tropo)
        check_t = staticmethod(gf.check_t)
        e_r_corr = staticmethod(gf.e_r_corr)
        cart2geo = staticmethod(gf.cart2geo)
        find_utm_zone = staticmethod(gf.findUtmZone)
        cart2utm = staticmethod(gf.cart2utm)

        @staticmethod
        def least_square_pos(sat, obs, c, trop):
            s = _S()
            s.useTropCorr = bool(trop)
            return gf.leastSquarePos(sat, obs, s)

    return [("oracle", orc), ("product", Product)]


@pytest.mark.parametrize("name,impl", _impls())
def test_position_solution_matches_reference(name, impl):
    g = load_golden("geo_cases.npz")
    for ci in range(g["eph"].shape[0]):
        prn = g["prn"][ci].astype(int)
        pos, clk = impl.satpos(float(g["tow"][ci]), prn, g["eph"][ci])
        assert np.max(np.abs(pos - g["sat_all"][ci])) < 1e-6
        assert np.max(np.abs(clk - g["clk_all"][ci])) < 1e-15
        vis = g["vis"][ci].astype(int)
        vis = vis[vis >= 0]
        obs = g["obs"][ci][:vis.size] + g["clk_all"][ci][vis] * _S.c
        p, el, az, dop = impl.least_square_pos(g["sat_all"][ci][:, vis], obs, _S.c, int(g["trop"][ci]))
        assert np.max(np.abs(np.asarray(p).reshape(-1) - g["pos"][ci])) < 1e-6
        assert np.max(np.abs(el - g["el"][ci][:vis.size])) < 1e-10 and np.max(np.abs(az - g["az"][ci][:vis.size])) < 1e-10
        assert np.max(np.abs(dop - g["dop"][ci])) < 1e-10
        assert np.linalg.norm(g["pos"][ci][:3] - g["rx"][ci]) < 100.0          # and it is a sensible fix
        lat, lon, h = impl.cart2geo(*g["pos"][ci][:3], 4)
        assert abs(lat - g["geo"][ci][0]) < 1e-11 and abs(lon - g["geo"][ci][1]) < 1e-11 and abs(h - g["geo"][ci][2]) < 1e-6
        zone = impl.find_utm_zone(g["geo"][ci][0], g["geo"][ci][1])
        assert zone == g["zone"][ci]
        assert np.max(np.abs(np.array(impl.cart2utm(*g["pos"][ci][:3], int(zone))) - g["utm"][ci])) < 1e-6


@pytest.mark.parametrize("name,impl", _impls())
def test_geo_helpers_match_reference(name, impl):
    g = load_golden("geo_cases.npz")
    t0 = [impl.tropo(v, 0.0, 1013.0, 293.0, 50.0, 0.0, 0.0, 0.0) for v in g["tropo_sinel"]]
    t1 = [impl.tropo(v, 1.2, 900.0, 280.0, 70.0, 1.0, 1.1, 1.3) for v in g["tropo_sinel"]]
    assert np.max(np.abs(np.array(t0) - g["tropo"])) < 1e-12 and np.max(np.abs(np.array(t1) - g["tropo_alt"])) < 1e-12
    assert np.array_equal([impl.check_t(v) for v in g["check_t_in"]], g["check_t"])
    xs = g["sat_all"][0][:, :6]
    erc = np.stack([impl.e_r_corr(0.066 + 0.004 * k, xs[:, k]) for k in range(6)])
    assert np.max(np.abs(erc - g["erc"])) < 1e-6
    tg = np.array([impl.togeod(6378137, 298.257223563, *p) for p in g["pts"]])
    assert np.max(np.abs(tg - g["togeod"])) < 1e-9
    tc = np.array([impl.topocent(g["pts"][k], xs[:, k % 6] - g["pts"][k]) for k in range(len(g["pts"]))])
    assert np.max(np.abs(tc[:, :2] - g["topocent"][:, :2])) < 1e-10 and np.max(np.abs(tc[:, 2] - g["topocent"][:, 2])) < 1e-6
    cg = np.array([[impl.cart2geo(p[0], p[1], p[2], i) for i in range(5)] for p in g["pts"][:6]])
    assert np.max(np.abs(cg[..., :2] - g["cart2geo"][..., :2])) < 1e-11 and np.max(np.abs(cg[..., 2] - g["cart2geo"][..., 2])) < 1e-6
    assert np.array_equal([impl.find_utm_zone(a, b) for a, b in g["zone_in"]], g["zone_out"])
    for bad in ((10.0, 181.0), (85.0, 10.0), (-81.0, 0.0)):
        with pytest.raises(IOError):
            impl.find_utm_zone(*bad)
    same = np.tile(g["sat_all"][0][:, :1], (1, 5))
    p0, el0, az0, dop0 = impl.least_square_pos(same, np.full(5, 2.2e7), _S.c, 1)
    assert np.asarray(p0).shape == tuple(g["deficient_shape"]) and np.abs(np.asarray(p0)).sum() + np.abs(dop0).sum() == 0.0


def rebuild_tracking(g, n_code=16368, ms=37000):
    """(PRN list, status list, absoluteSample rows, I_P rows) of the reference tracker's output, from the compact
    form stored in fix_scene.npz (block lengths, sign and RMS of I_P)."""
    blk = np.concatenate([g["first_block"][:, None], g["blk_offset"].astype(np.int64) + n_code], axis=1)
    abs_rows = np.cumsum(blk, axis=1).astype(np.float64)
    sign = np.unpackbits(g["ip_sign"], axis=1)[:, :ms].astype(np.float64) * 2 - 1
    ip_rows = sign * g["ip_rms"][:, None]
    return [int(p) for p in g["PRN"]], ['T'] * len(g["PRN"]), list(abs_rows), list(ip_rows)


def compare_solutions(got, g, tol_m=1e-6):
    n_meas = int(np.sum(np.isfinite(g["X"])))
    assert n_meas == 63
    for k in ("X", "Y", "Z", "dt", "height", "E", "N", "U"):
        assert np.max(np.abs(np.asarray(got[k], dtype=np.float64)[:n_meas] - g[k][:n_meas])) < tol_m, k
        assert np.all(np.isnan(np.asarray(got[k], dtype=np.float64)[n_meas:]))
    for k in ("latitude", "longitude"):
        assert np.max(np.abs(np.asarray(got[k], dtype=np.float64)[:n_meas] - g[k][:n_meas])) < 1e-10, k
    assert np.max(np.abs(np.asarray(got["DOP"], dtype=np.float64) - g["DOP"])) < 1e-9
    for k, name in (("rawP", "rawP"), ("correctedP", "correctedP"), ("el", "el"), ("az", "az")):
        a, b = np.asarray(got[k], dtype=np.float64), g[name]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.nanmax(np.abs(a - b)) < (1e-9 if k in ("el", "az") else tol_m), k
    assert float(got["utmZone"]) == float(g["utmZone"])


def test_oracle_post_navigate_matches_reference():
    """The whole navigation chain of the oracle on the reference tracker's output vs the reference's postNavigate."""
    g = load_golden("fix_scene.npz")
    prn, status, abs_rows, ip_rows = rebuild_tracking(g)
    so = orc.OracleSettings(samplingFreq=16368000.0, IF=4130400.0, numberOfChannels=len(prn), msToProcess=37000.0)
    out = orc.post_navigate(so, prn, status, abs_rows, ip_rows)
    assert np.array_equal(out["firstSubFrame"], g["firstSubFrame"])
    assert np.array_equal(out["eph"], g["eph"])
    assert np.array_equal(out["PRN"], g["chPRN"])
    compare_solutions(out, g)
    err = np.linalg.norm(np.stack([g["X"], g["Y"], g["Z"]])[:, :63] - g["rx"][:, None], axis=0)
    assert np.median(err) < 30.0 and err.max() < 80.0          # the reference finds the simulated receiver
