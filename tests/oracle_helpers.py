"""Oracle-side helpers shared by the GPU parity tests (and by tools/full_parity.py, which imports them from here:
tests never import from tools/)."""
import numpy as np

from oracle import softgnss_oracle as orc   # checker only


def oracle_channel(args):
    """One channel of `ms` code periods through the numpy restatement of the reference's track() (tracking.py:13-295).
    args = (host record, PRN, acquiredFreq, codePhase, ms); returns the 13 series, shape [13, ms].  Top-level, so that a
    process pool can run the channels side by side."""
    host, prn, freq, phase, ms = args
    so = orc.OracleSettings(numberOfChannels=1, msToProcess=float(ms))
    ch = dict(PRN=np.array([prn]), acquiredFreq=np.array([freq]), codePhase=np.array([phase]), status=np.array(['T']))
    return orc.stack_series(orc.track(so, ch, host))[0]
