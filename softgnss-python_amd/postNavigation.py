"""NavigationResult with the reference's interface, limited to the stage next to the accelerated path:
bit synchronisation and preamble search on the tracking output (reference postNavigation.py:443-631).

findPreambles(), navPartyChk(), the bit integration and calculatePseudoranges() are answered by libsgx.so
(sgx_find_preambles, sgx_nav_parity_check, sgx_nav_bits, sgx_pseudoranges); decodeEphemerides() runs the first half
of postNavigate (preambles -> bits -> sgx_ephemeris).  Satellite positions and the least-squares position
solution (the rest of postNavigate, plot) stay in the reference: they are scalar, millisecond-rate code outside this engine's scope.
"""
from __future__ import print_function

import ctypes as C

import numpy as np

from . import _native, engine
from .initialize import Result


class NavigationResult(Result):
    def __init__(self, trackResult, device=None):
        self._results = trackResult.results
        self._channels = trackResult.channels
        self._settings = trackResult.settings
        self._solutions = None
        self._eph = None
        self._device = device

    @staticmethod
    def navPartyChk(ndat):
        """Parity status of a GPS word (+1 / -1 passed, 0 failed); ndat = 32 values of +-1 (D29*, D30*, d1..d24,
        D25..D30), d1..d24 flipped in place when D30* != 1 (reference postNavigation.py:443-521)."""
        buf = np.ascontiguousarray(ndat, dtype=np.float64)
        st = C.c_int32(0)
        _native.check(_native.lib().sgx_nav_parity_check(buf.ctypes.data_as(C.c_void_p), C.byref(st)))
        ndat[...] = buf          # the reference mutates its argument
        return st.value

    def findPreambles(self):
        """(firstSubFrame, activeChnList) as in reference postNavigation.py:524-631."""
        assert isinstance(self._results, np.recarray)
        trackResults = self._results
        settings = self._settings
        firstSubFrame = np.zeros(settings.numberOfChannels, dtype=int)
        activeChnList = (trackResults.status != b'-').nonzero()[0] if trackResults.status.dtype.kind == "S" \
            else (trackResults.status != '-').nonzero()[0]
        n_act = len(activeChnList)
        if n_act:
            # quirk kept: row channelNr of the results serves the channelNr-th active channel
            i_p = np.stack([np.asarray(trackResults[k].I_P, dtype=np.float64) for k in range(n_act)])
            ctx = engine.get_context(settings, self._device)
            firstSubFrame[:n_act] = ctx.find_preambles(i_p, 0)
        for channelNr in range(n_act):
            if firstSubFrame[channelNr] == 0:
                activeChnList = np.setdiff1d(activeChnList, channelNr)
                print('Could not find valid preambles in channel %2d !' % channelNr)
        return firstSubFrame, activeChnList

    def navBits(self, subFrameStart, activeChnList):
        """{channelNr: list of '0'/'1'} - the navBitsBin the reference hands to ephemeris.ephemeris() as
        (navBitsBin[1:], navBitsBin[0]) for every active channel (reference postNavigation.py:125-138)."""
        out = {}
        for channelNr in activeChnList:
            bits = _native.nav_bits(np.asarray(self._results[channelNr].I_P, dtype=np.float64),
                                    int(subFrameStart[channelNr]))
            out[int(channelNr)] = [str(int(b)) for b in bits]
        return out

    def decodeEphemerides(self):
        """The first half of the reference's postNavigate (postNavigation.py:113-147): find the subframe starts,
        integrate the navigation bits and decode clock / orbit parameters and the time of week per channel.
        Returns (eph, TOW, subFrameStart, activeChnList); eph is the reference's recarray of 32 records with 27
        object fields, filled at index PRN-1.  Needs the 1500 bits after the first preamble, i.e. about 32 s."""
        from . import ephemeris as eph_mod
        subFrameStart, activeChnList = self.findPreambles()
        eph = np.recarray((32,), formats=['O'] * 27, names=','.join(eph_mod.FIELDS))
        TOW = None
        bits = self.navBits(subFrameStart, activeChnList)
        for channelNr in activeChnList:
            navBitsBin = bits[int(channelNr)]
            prn = int(self._results[channelNr].PRN)
            eph[prn - 1], TOW = eph_mod.ephemeris(navBitsBin[1:], navBitsBin[0])
            if eph[prn - 1].IODC is None or eph[prn - 1].IODE_sf2 is None or eph[prn - 1].IODE_sf3 is None:
                activeChnList = np.setdiff1d(activeChnList, channelNr)
        self._eph = eph
        return eph, TOW, subFrameStart, activeChnList

    def postNavigate(self):
        raise NotImplementedError("satellite positions and the least-squares fix stay in the reference "
                                  "(postNavigation.py:150-305, geoFunctions); decodeEphemerides() covers "
                                  "postNavigation.py:113-147")

    def calculatePseudoranges(self, msOfTheSignal, channelList):
        """Relative pseudoranges (metres, +inf for channels not in channelList) at millisecond
        msOfTheSignal[channel] of the tracking results (reference postNavigation.py:27-72)."""
        trackResults = self._results
        settings = self._settings
        abs_s = np.ascontiguousarray(np.stack([np.asarray(r.absoluteSample, dtype=np.float64) for r in trackResults]))
        when = np.ascontiguousarray(msOfTheSignal, dtype=np.float64)
        chans = np.ascontiguousarray(channelList, dtype=np.int32)
        if when.shape[0] < settings.numberOfChannels:
            raise IndexError("msOfTheSignal needs one entry per configured channel")
        out = np.empty(settings.numberOfChannels)
        rc = _native.lib().sgx_pseudoranges(abs_s.ctypes.data_as(C.c_void_p), abs_s.shape[0], abs_s.shape[1],
                                            when.ctypes.data_as(C.c_void_p), chans.ctypes.data_as(C.c_void_p),
                                            chans.shape[0], int(settings.numberOfChannels),
                                            int(settings.samplesPerCode), float(settings.startOffset),
                                            float(settings.c), out.ctypes.data_as(C.c_void_p))
        if rc == _native.SGX_E_RANGE:
            raise IndexError(_native.last_error())
        _native.check(rc)
        return out

    def plot(self):
        raise NotImplementedError("plotting is outside the accelerated path (reference postNavigation.py:307-439)")
