"""The functions of the reference's geoFunctions package that the position solution uses
(reference geoFunctions/__init__.py), answered by libsgx.so: same names, arguments and return values."""
import ctypes as C

import numpy as np

from . import _native
from .ephemeris import FIELDS as _EPH_FIELDS


def _d(n=1):
    return [C.c_double(0) for _ in range(n)]


def _vec(x, n):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d values" % n)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc == _native.SGX_E_RANGE:
        msg = _native.last_error()
        exc = IOError if msg.startswith("IOError") else (IndexError if msg.startswith("IndexError") else ValueError)
        raise exc(msg)
    _native.check(rc)


def check_t(time, *args, **kwargs):
    out = C.c_double(0)
    _check(_native.lib().sgx_check_t(float(time), C.byref(out)))
    return out.value


def e_r_corr(traveltime, X_sat, *args, **kwargs):
    out = np.zeros(3)
    _check(_native.lib().sgx_e_r_corr(float(traveltime), _p(_vec(X_sat, 3)), _p(out)))
    return out


def togeod(a, finv, X, Y, Z, *args, **kwargs):
    r = _d(3)
    _check(_native.lib().sgx_togeod(float(a), float(finv), float(X), float(Y), float(Z), *[C.byref(v) for v in r]))
    return r[0].value, r[1].value, r[2].value


def topocent(X, dx, *args, **kwargs):
    r = _d(3)
    _check(_native.lib().sgx_topocent(_p(_vec(X, 3)), _p(_vec(dx, 3)), *[C.byref(v) for v in r]))
    return r[0].value, r[1].value, r[2].value


def tropo(sinel, hsta, p, tkel, hum, hp, htkel, hhum):
    out = C.c_double(0)
    _check(_native.lib().sgx_tropo(float(sinel), float(hsta), float(p), float(tkel), float(hum), float(hp),
                                   float(htkel), float(hhum), C.byref(out)))
    return out.value


def eph_table(eph):
    """float64[32, 27] from the reference's eph recarray (records never filled stay zero)."""
    tab = np.zeros((32, len(_EPH_FIELDS)))
    for i in range(min(32, len(eph))):
        if eph[i][_EPH_FIELDS[0]] is None:
            continue
        tab[i] = [float(eph[i][k]) for k in _EPH_FIELDS]
    return tab


def satpos(transmitTime, prnList, eph, settings, *args, **kwargs):
    """(satPositions [3, n], satClkCorr [n]) at transmitTime for the PRNs in prnList."""
    prn = np.ascontiguousarray(np.asarray(prnList).reshape(-1), dtype=np.int32)
    tab = eph if isinstance(eph, np.ndarray) and eph.dtype == np.float64 else eph_table(eph)
    tab = np.ascontiguousarray(tab)
    pos = np.zeros((3, prn.size))
    clk = np.zeros(prn.size)
    _check(_native.lib().sgx_satpos(float(transmitTime), _p(prn), int(prn.size), _p(tab), _p(pos), _p(clk)))
    return pos, clk


def leastSquarePos(satpos_, obs, settings, *args, **kwargs):
    """(pos [X, Y, Z, dt], el, az, dop) - reference geoFunctions/__init__.py:636-739."""
    sp = np.ascontiguousarray(satpos_, dtype=np.float64)
    ob = np.ascontiguousarray(obs, dtype=np.float64)
    n = sp.shape[1]
    pos, el, az, dop = np.zeros(4), np.zeros(n), np.zeros(n), np.zeros(5)
    deficient = C.c_int32(0)
    rc = _native.lib().sgx_least_square_pos(_p(sp), _p(ob), int(n), float(settings.c), 1 if settings.useTropCorr else 0,
                                            _p(pos), _p(el), _p(az), _p(dop), C.byref(deficient))
    if rc == _native.SGX_E_RANGE:
        raise np.linalg.LinAlgError(_native.last_error())
    _native.check(rc)
    if deficient.value:
        return np.zeros((4, 1)), el, az, dop      # what the reference returns when matrix_rank(A) != 4
    return pos, el, az, dop


def cart2geo(X, Y, Z, i, *args, **kwargs):
    r = _d(3)
    _check(_native.lib().sgx_cart2geo(float(X), float(Y), float(Z), int(i), *[C.byref(v) for v in r]))
    return r[0].value, r[1].value, r[2].value


def findUtmZone(latitude, longitude, *args, **kwargs):
    z = C.c_int32(0)
    _check(_native.lib().sgx_find_utm_zone(float(latitude), float(longitude), C.byref(z)))
    return float(z.value)       # the reference returns np.fix(...) + 1, a float


def cart2utm(X, Y, Z, zone, *args, **kwargs):
    r = _d(3)
    _check(_native.lib().sgx_cart2utm(float(X), float(Y), float(Z), int(zone), *[C.byref(v) for v in r]))
    return r[0].value, r[1].value, r[2].value
