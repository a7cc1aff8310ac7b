"""ephemeris.ephemeris of the reference (ephemeris.py:60-195) answered by libsgx.so (sgx_ephemeris)."""
import ctypes as C

import numpy as np

from . import _native

FIELDS = ('weekNumber', 'accuracy', 'health', 'T_GD', 'IODC', 't_oc', 'a_f2', 'a_f1', 'a_f0', 'IODE_sf2', 'C_rs',
          'deltan', 'M_0', 'C_uc', 'e', 'C_us', 'sqrtA', 't_oe', 'C_ic', 'omega_0', 'C_is', 'i_0', 'C_rc', 'omega',
          'omegaDot', 'IODE_sf3', 'iDot')
_INT_FIELDS = (0, 1, 2, 4, 5, 9, 17, 25)     # the reference returns Python ints for these


def ephemeris(bits, d30star):
    """(eph, TOW): 27-tuple of clock / orbit parameters and the time of week (s) of the first of five subframes.
    bits: at least 1500 characters '0'/'1' (the first one is the first bit of a subframe); d30star: '0' or '1'."""
    if len(bits) < 1500:
        raise TypeError('The parameter BITS must contain 1500 bits!')
    if any([not isinstance(x, str) for x in bits]):
        raise TypeError('The parameter BITS must be a character array!')
    if not isinstance(d30star, str):
        raise TypeError('The parameter D30Star must be a char!')
    arr = np.array([1 if x == '1' else 0 for x in bits], dtype=np.uint8)
    out = np.zeros(27)
    tow = C.c_int64(0)
    rc = _native.lib().sgx_ephemeris(arr.ctypes.data_as(C.c_void_p), int(arr.size), 1 if d30star == '1' else 0,
                                     out.ctypes.data_as(C.c_void_p), C.byref(tow))
    if rc == _native.SGX_E_RANGE:
        raise UnboundLocalError(_native.last_error())
    _native.check(rc)
    eph = tuple(int(v) if i in _INT_FIELDS else float(v) for i, v in enumerate(out))
    return eph, int(tow.value)
