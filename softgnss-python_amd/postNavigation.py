"""NavigationResult with the reference's interface (reference postNavigation.py): what happens to the tracking
output after the accelerated path.

findPreambles(), navPartyChk(), the bit integration and calculatePseudoranges() are answered by libsgx.so
(sgx_find_preambles - the one device kernel of this stage -, sgx_nav_parity_check, sgx_nav_bits, sgx_pseudoranges);
decodeEphemerides() runs the first half of postNavigate (preambles -> bits -> sgx_ephemeris); postNavigate() the
whole chain down to the position fix (geoFunctions.py -> sgx_satpos, sgx_least_square_pos, ...).  Everything after
the preamble correlation is scalar, millisecond-rate host code.  Only plot() is left to the reference.
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
        """Navigation solutions of reference postNavigation.py:75-305: subframe starts, ephemerides, then every
        navSolPeriod ms pseudoranges -> satellite positions -> least-squares position -> geodetic / UTM.
        Results in self.solutions (the reference's nested recarray, 64 measurement columns) and self.ephemeris.
        Everything runs in libsgx.so: sgx_find_preambles, sgx_nav_bits, sgx_ephemeris, then sgx_post_navigate for the
        loop over measurement epochs (pseudoranges, satellite positions, least-squares fix, geodetic / UTM)."""
        trackResults = self._results
        settings = self._settings
        status = trackResults.status
        n_tracked = int(np.sum(status != (b'-' if status.dtype.kind == 'S' else '-')))
        if settings.msToProcess < 36000 or n_tracked < 4:
            print('Record is to short or too few satellites tracked. Exiting!')
            self._solutions = None
            self._eph = None
            return
        eph, TOW, subFrameStart, activeChnList = self.decodeEphemerides()
        if activeChnList.size == 0 or activeChnList.size < 4:
            print('Too few satellites with ephemeris data for position calculations. Exiting!')
            self._solutions = None
            self._eph = None
            return
        # the loop over measurement epochs (reference postNavigation.py:150-290) runs in libsgx (sgx_post_navigate); here
        # its flat arrays are put into the reference's nested record arrays
        nch = int(settings.numberOfChannels)
        n_meas = int(np.fix(settings.msToProcess - subFrameStart.max()) / settings.navSolPeriod)
        abs_s = np.ascontiguousarray(np.stack([np.asarray(r.absoluteSample, dtype=np.float64) for r in trackResults]))
        prn_of_row = np.ascontiguousarray([int(r.PRN) for r in trackResults], dtype=np.int32)
        sfs = np.ascontiguousarray(subFrameStart, dtype=np.float64)
        ready = np.ascontiguousarray(activeChnList, dtype=np.int32)
        from .geoFunctions import eph_table
        tab = np.ascontiguousarray(eph_table(eph))
        ch_PRN, ch_el, ch_az, ch_rawP, ch_corrP = (np.empty((nch, 64)) for _ in range(5))
        DOP = np.empty((5, 64))
        flat = np.empty((10, 64))
        zone = C.c_int32(0)
        short = np.zeros(64, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        rc = _native.lib().sgx_post_navigate(
            p(abs_s), int(abs_s.shape[0]), int(abs_s.shape[1]), p(prn_of_row), p(sfs), p(ready), int(ready.size), nch, p(tab),
            int(TOW), int(settings.samplesPerCode), float(settings.startOffset), float(settings.c),
            float(settings.navSolPeriod), float(settings.elevationMask), 1 if settings.useTropCorr else 0, n_meas,
            p(ch_PRN), p(ch_el), p(ch_az), p(ch_rawP), p(ch_corrP), p(DOP), p(flat), C.byref(zone), p(short))
        if rc == _native.SGX_E_RANGE:
            msg = _native.last_error()
            raise IndexError(msg) if msg.startswith('IndexError') else np.linalg.LinAlgError(msg)
        _native.check(rc)
        for m in np.flatnonzero(short[:max(n_meas, 0)]):
            print('   Measurement No. %d' % m + ': Not enough information for position solution.')
        channel = np.rec.array([(ch_PRN, ch_el, ch_az, ch_rawP, ch_corrP)], formats=['O'] * 5,
                               names='PRN,el,az,rawP,correctedP')
        navSolutions = np.rec.array([(channel, DOP) + tuple(flat[i] for i in range(7)) +
                                     (float(zone.value) if np.any(short[:max(n_meas, 0)] == 0) and n_meas > 0 else 0,) +
                                     tuple(flat[i] for i in range(7, 10))], formats=['O'] * 13,
                                    names='channel,DOP,X,Y,Z,dt,latitude,longitude,height,utmZone,E,N,U')
        self._solutions = navSolutions
        self._eph = eph
        return

    @property
    def solutions(self):
        assert isinstance(self._solutions, np.recarray)
        return self._solutions

    @property
    def ephemeris(self):
        assert isinstance(self._solutions, np.recarray)
        return self._eph

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
        """Coordinate variations around the mean fix, the fixes in the UTM plane and the satellites' azimuth /
        elevation (the panels of reference postNavigation.py:307-439).  Needs matplotlib; prints a notice and
        returns without it."""
        from .initialize import _pyplot
        plt = _pyplot("NavigationResult.plot")
        if plt is None:
            return
        if self._solutions is None:
            print('PLOTNAVIGATION: No navigation data to plot.')
            return
        sol = self._solutions[0]
        ref = [np.nanmean(sol.E), np.nanmean(sol.N), np.nanmean(sol.U)]
        plt.figure(300)
        plt.clf()
        ax = plt.subplot(2, 2, (1, 2))
        for series, r0 in zip((sol.E, sol.N, sol.U), ref):
            ax.plot(series - r0)
        ax.set_title('Coordinates variations in UTM system')
        ax.legend(['E', 'N', 'U'])
        ax.set_xlabel('Measurement period: %i ms' % self._settings.navSolPeriod)
        ax.set_ylabel('Variations (m)')
        ax = plt.subplot(2, 2, 3)
        ax.plot(sol.E - ref[0], sol.N - ref[1], '+')
        ax.set_title('Positions in UTM system')
        ax.set_xlabel('East (m)')
        ax.set_ylabel('North (m)')
        ax.axis('equal')
        ax = plt.subplot(2, 2, 4, projection='polar')
        ch = sol.channel[0]
        ax.plot(np.radians(ch.az.T), 90 - ch.el.T, '.')
        ax.set_theta_zero_location('N')
        ax.set_theta_direction(-1)
        ax.set_title('Sky plot (mean PDOP: %.2f)' % np.nanmean(np.where(sol.DOP[1] > 0, sol.DOP[1], np.nan)))
