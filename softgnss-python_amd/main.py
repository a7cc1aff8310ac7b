"""The reference's main.py (banner, probeData, postProcessing) for this engine:

    python -m softgnss-python_amd.main record.bin [--fs 38192000 --IF 9548000 --ms 37000 --channels 8 --skip 0]

Prints the channel table, the tracking time and, when the record is long enough (36 s, four satellites with
ephemerides), the mean position fix."""
from __future__ import print_function

import argparse

import numpy as np

from . import initialize


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("fileName")
    ap.add_argument("--fs", type=float, default=None, help="samplingFreq [Hz]")
    ap.add_argument("--IF", type=float, default=None, help="intermediate frequency [Hz]")
    ap.add_argument("--ms", type=float, default=None, help="msToProcess")
    ap.add_argument("--channels", type=int, default=None, help="numberOfChannels")
    ap.add_argument("--skip", type=int, default=None, help="skipNumberOfBytes")
    ap.add_argument("--no-probe", action="store_true", help="skip the raw-data statistics")
    a = ap.parse_args(argv)
    print('\nWelcome to:  softGNSS on MI355X\n')
    settings = initialize.Settings()
    settings.fileName = a.fileName
    for name, val in (("samplingFreq", a.fs), ("IF", a.IF), ("msToProcess", a.ms), ("numberOfChannels", a.channels),
                      ("skipNumberOfBytes", a.skip)):
        if val is not None:
            setattr(settings, name, val)
    if not a.no_probe:
        print('Probing data "%s"...' % settings.fileName)
        p = settings.probeData()
        if p is not None:
            k = int(np.argmax(p["Pxx"]))
            print('  %d Welch segments, spectral peak at %.3f MHz, samples within [%d, %d]'
                  % (p["segments"], p["f_MHz"][k], p["hist_edges"][np.flatnonzero(p["hist"])[0]],
                     p["hist_edges"][np.flatnonzero(p["hist"])[-1]] + 1))
    acq, trk, nav = settings.postProcessing()
    if nav is not None and nav._solutions is not None:
        sol = nav.solutions[0]
        ok = np.isfinite(sol.X)
        print('  %d position fixes; mean latitude %.6f deg, longitude %.6f deg, height %.1f m (UTM zone %d)'
              % (int(ok.sum()), np.nanmean(sol.latitude), np.nanmean(sol.longitude), np.nanmean(sol.height),
                 int(sol.utmZone)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
