"""CPU oracle for the GPS L1 C/A acquisition + tracking hot path.  TEST INFRASTRUCTURE ONLY.

This is a numpy restatement of the algorithm of perrysou/SoftGNSS-python (reference files
initialize.py, acquisition.py, tracking.py), written in this repo's own words, keeping the
reference's IEEE-754 operation order wherever a rounding function (ceil/floor/%/argmax)
follows.  Each function cites the reference lines it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / the timed CPU baseline.  The product path (softgnss-python_amd/)
never imports it and fails loudly when its HIP library is missing.

Pinning: the oracle is checked bit-for-bit against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py which imports a lib2to3-converted
throw-away copy of /root/reference in the build container) by tests/test_oracle_golden.py.
Third-party arithmetic (numpy.fft = pocketfft, np.sin/cos/linspace) is *called*, not restated,
with numpy 2.2.6 - the version the goldens were captured with.
"""
import numpy as np

NUM_SERIES = 13
SERIES = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
          "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt")


class OracleSettings(object):
    """Attribute bag with the reference's names and defaults (initialize.py:85-173)."""

    def __init__(self, **kw):
        self.msToProcess = 37000.0
        self.numberOfChannels = 8
        self.skipNumberOfBytes = 0
        self.dataType = 'int8'
        self.IF = 9548000.0
        self.samplingFreq = 38192000.0
        self.codeFreqBasis = 1023000.0
        self.codeLength = 1023
        self.acqSatelliteList = range(1, 33)
        self.acqSearchBand = 14.0
        self.acqThreshold = 2.5
        self.dllDampingRatio = 0.7
        self.dllNoiseBandwidth = 2.0
        self.dllCorrelatorSpacing = 0.5
        self.pllDampingRatio = 0.7
        self.pllNoiseBandwidth = 25.0
        self.c = 299792458.0          # initialize.py:170 (m/s)
        self.startOffset = 68.802     # initialize.py:172 (ms)
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def samplesPerCode(self):
        # initialize.py:183-185
        return int(np.round(self.samplingFreq / (self.codeFreqBasis / self.codeLength)))


G2_DELAY = (5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
            469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862)


def generate_ca_code(prn0):
    """float64[1023] of +-1.0 for PRN index 0..31.  Follows initialize.py:234-302.

    Two 10-stage registers of +-1 loaded with -1; output is the last stage; G1 feedback
    stage3*stage10, G2 feedback stage2*3*6*8*9*10; G2 rotated right by the PRN's delay;
    code = -(g1*g2).
    """
    if prn0 not in range(0, 32):
        raise AssertionError("prn index out of range")  # initialize.py:250
    g1 = np.zeros(1023)
    g2 = np.zeros(1023)
    ra = [-1.0] * 10
    rb = [-1.0] * 10
    for i in range(1023):
        g1[i] = ra[9]
        g2[i] = rb[9]
        fa = ra[2] * ra[9]
        fb = rb[1] * rb[2] * rb[5] * rb[7] * rb[8] * rb[9]
        ra = [fa] + ra[:9]
        rb = [fb] + rb[:9]
    d = G2_DELAY[prn0]
    g2 = np.concatenate((g2[1023 - d:], g2[:1023 - d]))
    return -g1 * g2


def ca_table_index(s):
    """Chip index of every sample of one code period.  Follows initialize.py:210-226 (A3).

    idx[k-1] = ceil((ts*k)/tc) - 1 for k = 1..N, last entry forced to 1022.  Multiply then
    divide, both rounded: the value sits exactly on a chip boundary every 112 samples and
    the fp64 result is what decides the index there.
    """
    n = s.samplesPerCode
    ts = 1.0 / s.samplingFreq
    tc = 1.0 / s.codeFreqBasis
    idx = np.ceil(ts * np.arange(1, n + 1) / tc) - 1
    idx = idx.astype(np.int64)
    idx[-1] = 1022
    return idx


def make_ca_table(s):
    """float64[32, N] sampled codes (initialize.py:188-231)."""
    idx = ca_table_index(s)
    out = np.zeros((32, s.samplesPerCode))
    for p in range(32):
        out[p] = generate_ca_code(p)[idx]
    return out


def calc_loop_coef(lbw, zeta, k):
    """(tau1, tau2) of the 2nd-order loop filter (initialize.py:304-328)."""
    wn = lbw * 8.0 * zeta / (4.0 * zeta ** 2 + 1)
    return k / (wn * wn), 2.0 * zeta / wn


def exclusion_index(code_phase, n, spc):
    """Index list searched for the second peak (acquisition.py:147-159, A8b).

    Kept quirks: a first-branch list that reaches index n raises IndexError when used
    (code_phase == spc, Q5); a second-branch list may start at -1 (wraps to n-1).
    """
    e1 = code_phase - spc
    e2 = code_phase + spc
    if e1 <= 0:
        return np.arange(e2, n + e1 + 1)
    if e2 >= n - 1:
        return np.arange(e2 - n, e1)
    return np.concatenate((np.arange(0, e1 + 1), np.arange(e2, n)))


def freq_bins(s):
    """Carrier grid (acquisition.py:68,99-101, A4)."""
    nb = int(np.round(s.acqSearchBand * 2) + 1)
    return np.array([s.IF - s.acqSearchBand / 2 * 1000 + 500.0 * k for k in range(nb)])


def acquire(s, long_signal, n_prn=None, n_blocks=2, noncoh=False, as_written=False, prn_indices=None):
    """Parallel code-phase search + fine carrier estimate.  Follows acquisition.py:49-203.

    n_prn      number of PRN indices searched (reference: len(acqSatelliteList), Q1).
    n_blocks   1-ms blocks used by the coarse search (reference: 2).
    noncoh     False = reference rule (keep the block with the larger maximum, A7);
               True  = EXTENSION beyond the reference (BASELINE.json config 4): rows are the
               sum of |corr|^2 over the n_blocks blocks.  This oracle is its definition.
    as_written True recomputes the PRN-independent carrier mix and forward FFTs for every
               PRN exactly as the reference does (same values; only used to time the
               reference's own cost).
    prn_indices  search only these PRN indices (0-based) instead of range(n_prn); the other entries stay zero
               (lets the timing harness spread the 32 independent searches over processes).
    Returns a dict with the three reference outputs plus the internal indices.
    """
    x = np.asarray(long_signal)
    n = s.samplesPerCode
    if n_prn is None:
        n_prn = len(s.acqSatelliteList)
    sig0dc = x - x.mean()                                   # acquisition.py:59
    ts = 1.0 / s.samplingFreq
    phase_points = np.arange(n) * 2 * np.pi * ts            # acquisition.py:65 (A2)
    bins = freq_bins(s)
    nb = len(bins)
    table = make_ca_table(s)
    blocks = [x[b * n:(b + 1) * n] for b in range(n_blocks)]

    def forward(k):
        sc = np.sin(bins[k] * phase_points)                 # acquisition.py:103-105 (A5)
        cc = np.cos(bins[k] * phase_points)
        return [np.fft.fft(sc * blk + 1j * (cc * blk)) for blk in blocks]

    fwd = None if as_written else [forward(k) for k in range(nb)]

    carr = np.zeros(32)
    cph = np.zeros(32)
    metric = np.zeros(32)
    fbin = np.full(32, -1, dtype=np.int64)
    fine = np.full(32, -1, dtype=np.int64)
    bsel = np.full((32, nb), -1, dtype=np.int64)
    spc = int(round(s.samplingFreq / s.codeFreqBasis))       # acquisition.py:145
    for p in (range(n_prn) if prn_indices is None else prn_indices):
        code_fd = np.fft.fft(table[p]).conj()               # acquisition.py:95
        res = np.zeros((nb, n))
        for k in range(nb):
            spec = forward(k) if as_written else fwd[k]
            pw = [abs(np.fft.ifft(sp * code_fd)) ** 2 for sp in spec]   # acquisition.py:120-126 (A6)
            if noncoh:
                acc = pw[0]
                for q in pw[1:]:
                    acc = acc + q
                res[k] = acc
                bsel[p, k] = 0
            else:
                # acquisition.py:129-133 (A7) generalised left-to-right: later block wins ties
                best = 0
                for b in range(1, n_blocks):
                    if not (pw[best].max() > pw[b].max()):
                        best = b
                res[k] = pw[best]
                bsel[p, k] = best
        fbi = int(res.max(1).argmax())                      # acquisition.py:139-140 (A8)
        peak = res.max(0).max()
        c = int(res.max(0).argmax())                        # acquisition.py:142-143
        rng = exclusion_index(c, n, spc)
        second = res[fbi, rng].max()                        # acquisition.py:162 (IndexError kept)
        metric[p] = peak / second
        fbin[p] = fbi
        if peak / second > s.acqThreshold:                  # acquisition.py:166
            code = generate_ca_code(p)
            cvi = np.floor(ts * np.arange(1, 10 * n + 1) / (1.0 / s.codeFreqBasis))   # :172 (A9)
            long_code = code[(cvi % 1023).astype(np.int64)]
            xc = sig0dc[c:c + 10 * n] * long_code           # acquisition.py:177
            npts = int(8 * 2 ** (np.ceil(np.log2(len(xc)))))
            mag = np.abs(np.fft.fft(xc, npts))              # acquisition.py:182
            uniq = int(np.ceil((npts + 1) / 2.0))
            m = int(mag[4:uniq - 5].argmax())               # acquisition.py:187 (A10)
            carr[p] = (np.arange(uniq) * s.samplingFreq / npts)[m]   # index not re-offset (Q3)
            cph[p] = c
            fine[p] = m
    return dict(carrFreq=carr, codePhase=cph, peakMetric=metric, freqBin=fbin, fineIdx=fine, blockSel=bsel)


def pre_run(s, acq):
    """Channel table from acquisition results.  Follows acquisition.py:259-306 (a9).

    Stable descending sort on peakMetric; the first min(numberOfChannels, #carrFreq>0) entries
    become channels (PRN 1-based, 0 = off).
    """
    nch = int(s.numberOfChannels)
    prn = np.zeros(nch, dtype=np.int64)
    freq = np.zeros(nch)
    cph = np.zeros(nch)
    status = ['-'] * nch
    order = sorted(enumerate(acq["peakMetric"]), key=lambda t: t[-1], reverse=True)
    for i in range(min(nch, int(np.sum(acq["carrFreq"] > 0)))):
        j = order[i][0]
        prn[i] = j + 1
        freq[i] = acq["carrFreq"][j]
        cph[i] = acq["codePhase"][j]
        status[i] = 'T'
    return dict(PRN=prn, acquiredFreq=freq, codePhase=cph, status=status)


def track(s, channels, record, ms=None):
    """DLL/PLL tracking of every active channel.  Follows tracking.py:35-294.

    record   array holding the file from byte 0 (the reference seeks/reads a file object; positions
             here are the same BYTE offsets).  Samples are s.dataType (tracking.py:154 reads
             np.fromfile(fid, settings.dataType, blksize)); note that the reference seeks to
             skipNumberOfBytes + codePhase BYTES and records fid.tell() in bytes for every dtype
             (tracking.py:107,255) - kept as it is.
    Returns None on a short read (tracking.py:159-163), else a list with one dict per ACTIVE
    channel (Q8) holding PRN, status and the 13 per-ms float64 series.
    """
    dt = np.dtype(getattr(s, "dataType", "int8"))
    isz = dt.itemsize
    rec = np.ascontiguousarray(record).view(np.uint8).ravel()      # the file's bytes
    n_ms = int(s.msToProcess if ms is None else ms)
    spc_el = s.dllCorrelatorSpacing
    pdi = 0.001
    t1c, t2c = calc_loop_coef(s.dllNoiseBandwidth, s.dllDampingRatio, 1.0)     # tracking.py:45
    t1p, t2p = calc_loop_coef(s.pllNoiseBandwidth, s.pllDampingRatio, 0.25)    # tracking.py:52
    fs = s.samplingFreq
    out = []
    for ch in range(int(s.numberOfChannels)):
        if channels["PRN"][ch] == 0:
            continue
        ser = {k: (np.zeros(n_ms) if k in ("absoluteSample", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L")
                   else np.inf * np.ones(n_ms)) for k in SERIES}       # tracking.py:65-94
        pos = int(s.skipNumberOfBytes + channels["codePhase"][ch])     # tracking.py:107
        code = generate_ca_code(int(channels["PRN"][ch]) - 1)
        code = np.r_[code[-1], code, code[0]]                          # tracking.py:111
        code_freq = s.codeFreqBasis
        rem_code = 0.0
        carr_freq = channels["acquiredFreq"][ch]
        carr_basis = channels["acquiredFreq"][ch]
        rem_carr = 0.0
        old_code_nco = old_code_err = old_carr_nco = old_carr_err = 0.0
        for it in range(n_ms):
            step = code_freq / fs                                       # tracking.py:148 (T1)
            blk = int(np.ceil((s.codeLength - rem_code) / step))
            chunk = rec[pos:pos + blk * isz].tobytes()
            raw = np.frombuffer(chunk[:len(chunk) - len(chunk) % isz], dtype=dt)   # tracking.py:154 (T2): whole items
            if len(raw) != blk:
                return None
            pos += blk * isz
            # tracking.py:166-188 (T3): three linspace ramps, ceil, gather
            te = np.linspace(rem_code - spc_el, blk * step + rem_code - spc_el, blk, endpoint=False)
            early = code[np.ceil(te).astype(np.int64)]
            tl = np.linspace(rem_code + spc_el, blk * step + rem_code + spc_el, blk, endpoint=False)
            late = code[np.ceil(tl).astype(np.int64)]
            tp = np.linspace(rem_code, blk * step + rem_code, blk, endpoint=False)
            prompt = code[np.ceil(tp).astype(np.int64)]
            rem_code = tp[blk - 1] + step - 1023.0                      # tracking.py:190 (T4)
            tm = np.arange(0, blk + 1) / fs                             # tracking.py:193 (T5)
            arg = carr_freq * 2.0 * np.pi * tm + rem_carr
            rem_carr = arg[blk] % (2 * np.pi)
            ccos = np.cos(arg[0:blk])
            csin = np.sin(arg[0:blk])
            qbb = ccos * raw                                            # tracking.py:205-219 (T6)
            ibb = csin * raw
            i_e = (early * ibb).sum()
            q_e = (early * qbb).sum()
            i_p = (prompt * ibb).sum()
            q_p = (prompt * qbb).sum()
            i_l = (late * ibb).sum()
            q_l = (late * qbb).sum()
            with np.errstate(divide="ignore", invalid="ignore"):
                carr_err = np.arctan(q_p / i_p) / 2.0 / np.pi           # tracking.py:223 (T7)
            carr_nco = old_carr_nco + t2p / t1p * (carr_err - old_carr_err) + carr_err * (pdi / t1p)
            old_carr_nco = carr_nco
            old_carr_err = carr_err
            carr_freq = carr_basis + carr_nco
            ee = np.sqrt(i_e * i_e + q_e * q_e)                         # tracking.py:238 (T8)
            ll = np.sqrt(i_l * i_l + q_l * q_l)
            code_err = (ee - ll) / (ee + ll)
            code_nco = old_code_nco + t2c / t1c * (code_err - old_code_err) + code_err * (pdi / t1c)
            old_code_nco = code_nco
            old_code_err = code_err
            code_freq = s.codeFreqBasis - code_nco
            ser["absoluteSample"][it] = pos                             # tracking.py:255 (T9)
            ser["codeFreq"][it] = code_freq
            ser["carrFreq"][it] = carr_freq
            ser["I_P"][it] = i_p
            ser["I_E"][it] = i_e
            ser["I_L"][it] = i_l
            ser["Q_E"][it] = q_e
            ser["Q_P"][it] = q_p
            ser["Q_L"][it] = q_l
            ser["dllDiscr"][it] = code_err
            ser["dllDiscrFilt"][it] = code_nco
            ser["pllDiscr"][it] = carr_err
            ser["pllDiscrFilt"][it] = carr_nco
        ser["PRN"] = int(channels["PRN"][ch])
        ser["status"] = channels["status"][ch]
        out.append(ser)
    return out


def stack_series(tracks):
    """[n_active, 13, ms] float64 view of track() output in SERIES order."""
    return np.stack([np.stack([t[k] for k in SERIES]) for t in tracks])


# ---- next row: bit sync + preamble search on the tracking output ----------

PREAMBLE_BITS = (1, -1, -1, -1, 1, -1, 1, 1)     # postNavigation.py:552


def nav_party_chk(ndat):
    """Parity status of one GPS word: +1 / -1 (passed, data polarity) or 0 (failed).
    Follows postNavigation.py:443-521 (ICD-200C table 20-XIV with +-1 for bits, products for XOR).
    ndat: 32 values of +-1 = D29*, D30*, d1..d24, D25..D30; like the reference it flips d1..d24 IN PLACE
    when D30* is not +1."""
    if ndat[1] != 1:
        ndat[2:26] *= (-1)
    rows = ((0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24),
            (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25),
            (0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23),
            (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24),
            (1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25),
            (0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25))
    parity = np.array([np.prod([ndat[i] for i in r]) for r in rows])
    if (parity == ndat[26:]).sum() == 6:
        return -1 * ndat[1]
    return 0


def preamble_correlation(i_p, search_start=0):
    """c[t] = sum_k sign(I_P)[t+k] * preamble_ms[k] for t = 0..L-1, the window running off the end.
    Equals np.correlate(bits, zero-padded preamble, 'full')[L-1:2L] of postNavigation.py:573-583, computed
    with 160 taps instead of L."""
    bits = np.asarray(i_p, dtype=np.float64)[search_start:].copy()
    bits[bits > 0] = 1
    bits[bits <= 0] = -1
    pre = np.kron(np.array(PREAMBLE_BITS, dtype=np.float64), np.ones(20))
    return np.correlate(np.r_[bits, np.zeros(pre.size - 1)], pre, mode='valid')


def find_preambles(i_p_rows, statuses, n_channels, search_start=0):
    """First subframe start (ms index) per channel and the channels that have one.
    Follows postNavigation.py:524-631: |correlation| > 153, a second candidate exactly 6000 ms later, parity
    of the TLM and HOW words on 20-ms sums.  Quirk kept: row `channelNr` of the results is used for the
    channelNr-th active channel."""
    first = np.zeros(n_channels, dtype=int)
    active = np.array([i for i, st in enumerate(statuses) if st != '-'], dtype=int)
    for ch in range(len(active)):
        ip = np.asarray(i_p_rows[ch], dtype=np.float64)
        c = preamble_correlation(ip, search_start)
        index = (np.abs(c) > 153).nonzero()[0] + search_start
        for i in range(len(index)):
            if ((index - index[i]) == 6000).any():
                bits = ip[index[i] - 40:index[i] + 20 * 60].copy()
                bits = bits.reshape(20, -1, order='F').sum(0)     # ValueError like the reference if cut short
                bits[bits > 0] = 1
                bits[bits <= 0] = -1
                if nav_party_chk(bits[:32]) != 0 and nav_party_chk(bits[30:62]) != 0:
                    first[ch] = index[i]
                    break
        if first[ch] == 0:
            active = np.setdiff1d(active, ch)
    return first, active


def nav_bits(i_p, sub_frame_start):
    """postNavigation.py:125-138: the navBits array (0/1 ints) of one channel, same numpy calls as the reference."""
    samples = np.asarray(i_p, dtype=np.float64)[sub_frame_start - 20: sub_frame_start + 1500 * 20].copy()
    samples = samples.reshape(20, -1, order='F')
    return (samples.sum(0) > 0) * 1


# ---- next row: probeData statistics (reference initialize.py:330-417) -------------------------------------------
PROBE_NPERSEG = 16384
PROBE_NOVERLAP = 1024


def probe_stats(s, data):
    """The numbers Settings.probeData() plots for `data` = the first 10 code periods (int8):
    (f [MHz], Pxx) = welch(data - mean(data), fs/1e6, hamming(16384, sym=False), 16384, 1024, 16384) restated
    step by step (scipy 1.15 legacy `_spectral_helper`: constant detrend per segment, window, rfft, density
    scaling, one-sided doubling, mean over segments), and hist = np.histogram(data, arange(-128, 128))[0]
    (initialize.py:389-400).  numpy's rfft and scipy's window are called, not restated."""
    data = np.asarray(data)
    fs = s.samplingFreq / 1000000.0
    x = data - np.mean(data)
    nseg_len, nov = PROBE_NPERSEG, PROBE_NOVERLAP
    from scipy.signal.windows import hamming                         # third-party, called like the reference does
    win = hamming(nseg_len, False)                                   # 0.54 - 0.46 cos(2 pi k / 16384)
    step = nseg_len - nov
    nseg = (x.shape[-1] - nov) // step
    seg = np.stack([x[i * step:i * step + nseg_len] for i in range(nseg)])
    seg = seg - np.mean(seg, axis=-1, keepdims=True)                # detrend='constant'
    seg = win * seg
    spec = np.fft.rfft(seg, n=nseg_len)
    pxx = (np.conjugate(spec) * spec).real
    pxx *= 1.0 / (fs * (win * win).sum())                            # scaling='density'
    pxx[..., 1:-1] *= 2                                              # one-sided, even nfft
    pxx = pxx.T.mean(axis=-1)                                        # average='mean' over the segments
    f = np.fft.rfftfreq(nseg_len, 1 / fs)
    hist = np.histogram(data, np.arange(-128, 128))[0]
    return f, pxx, hist


def calculate_pseudoranges(s, absolute_sample_rows, ms_of_the_signal, channel_list):
    """postNavigation.py:27-72: relative pseudoranges in metres; absolute_sample_rows[i] = absoluteSample series of
    the i-th tracking record."""
    travel = np.inf * np.ones(s.numberOfChannels)
    n = s.samplesPerCode
    for ch in channel_list:
        travel[ch] = absolute_sample_rows[ch][int(ms_of_the_signal[ch])] / n
    minimum = np.floor(travel.min())
    travel = travel - minimum + s.startOffset
    return travel * s.c / 1000


def _u(sub, *slices):
    return int(''.join(''.join(sub[a:b]) for a, b in slices), 2)


def _s(sub, *slices):
    txt = ''.join(''.join(sub[a:b]) for a, b in slices)
    v = int(txt, 2)
    return v - 2 ** len(txt) if txt[0] == '1' else v            # ephemeris.py:7-25


def ephemeris(bits, d30star):
    """ephemeris.py:60-195: (eph 27-tuple, TOW) from 1500 bits ('0'/'1' characters) and the bit before them."""
    if len(bits) < 1500:
        raise TypeError('The parameter BITS must contain 1500 bits!')
    gps_pi = 3.1415926535898
    f = {}
    sub = None
    for i in range(5):
        sub = list(bits[300 * i:300 * (i + 1)])
        for j in range(10):
            if d30star == '1':                                      # checkPhase, ephemeris.py:30-57
                for k in range(30 * j, 30 * j + 24):
                    sub[k] = '0' if sub[k] == '1' else '1'
            d30star = sub[30 * (j + 1) - 1]
        sid = _u(sub, (49, 52))
        if sid == 1:
            f.update(weekNumber=_u(sub, (60, 70)) + 1024, accuracy=_u(sub, (72, 76)), health=_u(sub, (76, 82)),
                     T_GD=_s(sub, (195, 204)) * 2 ** (-31), IODC=_u(sub, (82, 84), (196, 204)),
                     t_oc=_u(sub, (218, 234)) * 2 ** 4, a_f2=_s(sub, (240, 248)) * 2 ** (-55),
                     a_f1=_s(sub, (248, 264)) * 2 ** (-43), a_f0=_s(sub, (270, 292)) * 2 ** (-31))
        elif sid == 2:
            f.update(IODE_sf2=_u(sub, (60, 68)), C_rs=_s(sub, (68, 84)) * 2 ** (-5),
                     deltan=_s(sub, (90, 106)) * 2 ** (-43) * gps_pi,
                     M_0=_s(sub, (106, 114), (120, 144)) * 2 ** (-31) * gps_pi, C_uc=_s(sub, (150, 166)) * 2 ** (-29),
                     e=_u(sub, (166, 174), (180, 204)) * 2 ** (-33), C_us=_s(sub, (210, 226)) * 2 ** (-29),
                     sqrtA=_u(sub, (226, 234), (240, 264)) * 2 ** (-19), t_oe=_u(sub, (270, 286)) * 2 ** 4)
        elif sid == 3:
            f.update(C_ic=_s(sub, (60, 76)) * 2 ** (-29), omega_0=_s(sub, (76, 84), (90, 114)) * 2 ** (-31) * gps_pi,
                     C_is=_s(sub, (120, 136)) * 2 ** (-29), i_0=_s(sub, (136, 144), (150, 174)) * 2 ** (-31) * gps_pi,
                     C_rc=_s(sub, (180, 196)) * 2 ** (-5), omega=_s(sub, (196, 204), (210, 234)) * 2 ** (-31) * gps_pi,
                     omegaDot=_s(sub, (240, 264)) * 2 ** (-43) * gps_pi, IODE_sf3=_u(sub, (270, 278)),
                     iDot=_s(sub, (278, 292)) * 2 ** (-43) * gps_pi)
    tow = _u(sub, (30, 47)) * 6 - 30
    names = ('weekNumber', 'accuracy', 'health', 'T_GD', 'IODC', 't_oc', 'a_f2', 'a_f1', 'a_f0', 'IODE_sf2', 'C_rs',
             'deltan', 'M_0', 'C_uc', 'e', 'C_us', 'sqrtA', 't_oe', 'C_ic', 'omega_0', 'C_is', 'i_0', 'C_rc', 'omega',
             'omegaDot', 'IODE_sf3', 'iDot')
    missing = [k for k in names if k not in f]
    if missing:
        raise UnboundLocalError("local variable '%s' referenced before assignment" % missing[0])
    return tuple(f[k] for k in names), tow


# ---- next row: satellite positions and least-squares fix (reference geoFunctions/__init__.py) -------------------
EPH_NAMES = ('weekNumber', 'accuracy', 'health', 'T_GD', 'IODC', 't_oc', 'a_f2', 'a_f1', 'a_f0', 'IODE_sf2', 'C_rs',
             'deltan', 'M_0', 'C_uc', 'e', 'C_us', 'sqrtA', 't_oe', 'C_ic', 'omega_0', 'C_is', 'i_0', 'C_rc', 'omega',
             'omegaDot', 'IODE_sf3', 'iDot')


def check_t(t):
    """geoFunctions/__init__.py:745-771: fold a time difference into +-half a week."""
    half = 302400.0
    if t > half:
        return t - 2 * half
    if t < -half:
        return t + 2 * half
    return t


def e_r_corr(traveltime, x_sat):
    """geoFunctions/__init__.py:491-523."""
    w = 7.292115147e-05 * traveltime
    r3 = np.array([[np.cos(w), np.sin(w), 0.0], [-np.sin(w), np.cos(w), 0.0], [0.0, 0.0, 1.0]])
    return r3.dot(x_sat)


def togeod(a, finv, X, Y, Z):
    """geoFunctions/__init__.py:892-996: (latitude deg, longitude deg, height)."""
    rtd = 180 / np.pi
    esq = 0.0 if finv < 1e-20 else (2 - 1 / finv) / finv
    oneesq = 1 - esq
    P = np.sqrt(X ** 2 + Y ** 2)
    dlambda = np.arctan2(Y, X) * rtd if P > 1e-20 else 0.0
    if dlambda < 0:
        dlambda = dlambda + 360
    r = np.sqrt(P ** 2 + Z ** 2)
    sinphi = Z / r if r > 1e-20 else 0.0
    dphi = np.arcsin(sinphi)
    if r < 1e-20:
        return dphi, dlambda, 0.0
    h = r - a * (1 - sinphi * sinphi / finv)
    for _ in range(10):
        sinphi = np.sin(dphi)
        cosphi = np.cos(dphi)
        n_phi = a / np.sqrt(1 - esq * sinphi * sinphi)
        dP = P - (n_phi + h) * cosphi
        dZ = Z - (n_phi * oneesq + h) * sinphi
        h = h + sinphi * dZ + cosphi * dP
        dphi = dphi + (cosphi * dZ - sinphi * dP) / (n_phi + h)
        if (dP * dP + dZ * dZ) < 1e-10:
            break
    return dphi * rtd, dlambda, h


def topocent(X, dx):
    """geoFunctions/__init__.py:1003-1064: (Az deg, El deg, distance)."""
    dtr = np.pi / 180
    phi, lam, _ = togeod(6378137, 298.257223563, X[0], X[1], X[2])
    cl, sl = np.cos(lam * dtr), np.sin(lam * dtr)
    cb, sb = np.cos(phi * dtr), np.sin(phi * dtr)
    F = np.array([[-sl, -sb * cl, cb * cl], [cl, -sb * sl, cb * sl], [0.0, cb, sb]])
    E, N, U = F.T.dot(dx)
    hor = np.sqrt(E ** 2 + N ** 2)
    if hor < 1e-20:
        az, el = 0.0, 90.0
    else:
        az = np.arctan2(E, N) / dtr
        el = np.arctan2(U, hor) / dtr
    if az < 0:
        az = az + 360
    return az, el, np.sqrt(dx[0] ** 2 + dx[1] ** 2 + dx[2] ** 2)


def tropo(sinel, hsta, p, tkel, hum, hp, htkel, hhum):
    """geoFunctions/__init__.py:1071-1186 (Goad & Goodman): tropospheric range correction, metres."""
    a_e, b0, tlapse = 6378.137, 7.839257e-05, -6.5
    tkhum = tkel + tlapse * (hhum - htkel)
    atkel = 7.5 * (tkhum - 273.15) / (237.3 + tkhum - 273.15)
    e0 = 0.0611 * hum * 10 ** atkel
    tksea = tkel - tlapse * htkel
    em = -978.77 / (2870400.0 * tlapse * 1e-05)
    e0sea = e0 * (tksea / (tksea + tlapse * hhum)) ** (4 * em)
    psea = p * (tksea / (tksea + tlapse * hp)) ** em
    if sinel < 0:
        sinel = 0
    total = 0.0
    refsea = 7.7624e-05 / tksea
    htop = 1.1385e-05 / refsea
    refsea = refsea * psea
    ref = refsea * ((htop - hsta) / htop) ** 4
    for second in (False, True):
        rtop = (a_e + htop) ** 2 - (a_e + hsta) ** 2 * (1 - sinel ** 2)
        if rtop < 0:
            rtop = 0
        rtop = np.sqrt(rtop) - (a_e + hsta) * sinel
        a = -sinel / (htop - hsta)
        b = -b0 * (1 - sinel ** 2) / (htop - hsta)
        rn = np.array([rtop ** (i + 2) for i in range(8)])
        alpha = np.array([2 * a, 2 * a ** 2 + 4 * b / 3, a * (a ** 2 + 3 * b),
                          a ** 4 / 5 + 2.4 * a ** 2 * b + 1.2 * b ** 2, 2 * a * b * (a ** 2 + 3 * b) / 3,
                          b ** 2 * (6 * a ** 2 + 4 * b) * 0.1428571, 0, 0])
        if b ** 2 > 1e-35:
            alpha[6] = a * b ** 3 / 2
            alpha[7] = b ** 4 / 9
        total += (rtop + alpha.dot(rn)) * ref * 1000
        if second:
            break
        refsea = (0.3719 / tksea - 1.292e-05) / tksea
        htop = 1.1385e-05 * (1255.0 / tksea + 0.05) / refsea
        ref = refsea * e0sea * ((htop - hsta) / htop) ** 4
    return total


def satpos(transmit_time, prn_list, eph_table):
    """geoFunctions/__init__.py:779-885.  eph_table: float [32][27] (EPH_NAMES order) -> (positions [3, n], clock s)."""
    gps_pi = 3.14159265359
    omegae_dot, GM, F = 7.2921151467e-05, 3.986005e+14, -4.442807633e-10
    n = len(prn_list)
    clk = np.zeros(n)
    pos = np.zeros((3, n))
    for k in range(n):
        q = dict(zip(EPH_NAMES, [float(v) for v in eph_table[int(prn_list[k]) - 1]]))
        dt = check_t(transmit_time - q['t_oc'])
        clk[k] = (q['a_f2'] * dt + q['a_f1']) * dt + q['a_f0'] - q['T_GD']
        time = transmit_time - clk[k]
        a = q['sqrtA'] * q['sqrtA']
        tk = check_t(time - q['t_oe'])
        n0 = np.sqrt(GM / a ** 3)
        M = q['M_0'] + (n0 + q['deltan']) * tk
        M = np.remainder(M + 2 * gps_pi, 2 * gps_pi)
        E = M
        for _ in range(10):
            E_old = E
            E = M + q['e'] * np.sin(E)
            if abs(np.remainder(E - E_old, 2 * gps_pi)) < 1e-12:
                break
        E = np.remainder(E + 2 * gps_pi, 2 * gps_pi)
        dtr = F * q['e'] * q['sqrtA'] * np.sin(E)
        nu = np.arctan2(np.sqrt(1 - q['e'] ** 2) * np.sin(E), np.cos(E) - q['e'])
        phi = np.remainder(nu + q['omega'], 2 * gps_pi)
        u = phi + q['C_uc'] * np.cos(2 * phi) + q['C_us'] * np.sin(2 * phi)
        r = a * (1 - q['e'] * np.cos(E)) + q['C_rc'] * np.cos(2 * phi) + q['C_rs'] * np.sin(2 * phi)
        i = q['i_0'] + q['iDot'] * tk + q['C_ic'] * np.cos(2 * phi) + q['C_is'] * np.sin(2 * phi)
        Om = q['omega_0'] + (q['omegaDot'] - omegae_dot) * tk - omegae_dot * q['t_oe']
        Om = np.remainder(Om + 2 * gps_pi, 2 * gps_pi)
        pos[0, k] = np.cos(u) * r * np.cos(Om) - np.sin(u) * r * np.cos(i) * np.sin(Om)
        pos[1, k] = np.cos(u) * r * np.sin(Om) + np.sin(u) * r * np.cos(i) * np.cos(Om)
        pos[2, k] = np.sin(u) * r * np.sin(i)
        clk[k] = (q['a_f2'] * dt + q['a_f1']) * dt + q['a_f0'] - q['T_GD'] + dtr
    return pos, clk


def least_square_pos(sat, obs, c, use_trop):
    """geoFunctions/__init__.py:636-739 -> (pos, el, az, dop); pos is zeros((4, 1)) for rank-deficient geometry."""
    dtr = np.pi / 180
    pos = np.zeros(4)
    n = sat.shape[1]
    A = np.zeros((n, 4))
    omc = np.zeros(n)
    az, el, dop = np.zeros(n), np.zeros(n), np.zeros(5)
    for it in range(7):
        for i in range(n):
            if it == 0:
                rot = sat[:, i].copy()
                trop = 2
            else:
                rho2 = (sat[0, i] - pos[0]) ** 2 + (sat[1, i] - pos[1]) ** 2 + (sat[2, i] - pos[2]) ** 2
                rot = e_r_corr(np.sqrt(rho2) / c, sat[:, i])
                az[i], el[i], _ = topocent(pos[0:3], rot - pos[0:3])
                trop = tropo(np.sin(el[i] * dtr), 0.0, 1013.0, 293.0, 50.0, 0.0, 0.0, 0.0) if use_trop else 0
            omc[i] = obs[i] - np.linalg.norm(rot - pos[0:3]) - pos[3] - trop
            A[i, :] = np.array([-(rot[0] - pos[0]) / obs[i], -(rot[1] - pos[1]) / obs[i], -(rot[2] - pos[2]) / obs[i], 1])
        if np.linalg.matrix_rank(A) != 4:
            return np.zeros((4, 1)), el, az, dop
        pos = pos + np.linalg.lstsq(A, omc, rcond=None)[0].flatten()
    Q = np.linalg.inv(A.T.dot(A))
    dop[:] = [np.sqrt(np.trace(Q)), np.sqrt(Q[0, 0] + Q[1, 1] + Q[2, 2]), np.sqrt(Q[0, 0] + Q[1, 1]), np.sqrt(Q[2, 2]),
              np.sqrt(Q[3, 3])]
    return pos, el, az, dop


def cart2geo(X, Y, Z, i):
    """geoFunctions/__init__.py:7-77 -> (phi deg, lambda deg, h)."""
    a = [6378388.0, 6378160.0, 6378135.0, 6378137.0, 6378137.0][i]
    f = [1 / 297, 1 / 298.247, 1 / 298.26, 1 / 298.257222101, 1 / 298.257223563][i]
    lam = np.arctan2(Y, X)
    ex2 = (2 - f) * f / ((1 - f) ** 2)
    c = a * np.sqrt(1 + ex2)
    phi = np.arctan(Z / (np.sqrt(X ** 2 + Y ** 2) * (1 - (2 - f)) * f))
    h, oldh, it = 0.1, 0, 0
    while abs(h - oldh) > 1e-12:
        oldh = h
        N = c / np.sqrt(1 + ex2 * np.cos(phi) ** 2)
        phi = np.arctan(Z / (np.sqrt(X ** 2 + Y ** 2) * (1 - (2 - f) * f * N / (N + h))))
        h = np.sqrt(X ** 2 + Y ** 2) / np.cos(phi) - N
        it += 1
        if it > 100:
            break
    return phi * (180 / np.pi), lam * (180 / np.pi), h


def find_utm_zone(latitude, longitude):
    """geoFunctions/__init__.py:529-571."""
    if longitude > 180 or longitude < -180:
        raise IOError('Longitude value exceeds limits (-180:180).')
    if latitude > 84 or latitude < -80:
        raise IOError('Latitude value exceeds limits (-80:84).')
    zone = np.fix((180 + longitude) / 6) + 1
    if latitude > 72:
        for lo, hi, z in ((0, 9, 31), (9, 21, 33), (21, 33, 35), (33, 42, 37)):
            if lo <= longitude < hi:
                zone = z
    elif 56 <= latitude < 64 and 3 <= longitude < 12:
        zone = 32
    return zone


def _clsin(ar, degree, argument):
    cos_arg = 2 * np.cos(argument)
    hr1 = hr = 0
    for t in range(degree, 0, -1):
        hr2, hr1 = hr1, hr
        hr = ar[t - 1] + cos_arg * hr1 - hr2
    return hr * np.sin(argument)


def _clksin(ar, degree, arg_real, arg_imag):
    sr, cr = np.sin(arg_real), np.cos(arg_real)
    shi, chi = np.sinh(arg_imag), np.cosh(arg_imag)
    r = 2 * cr * chi
    i = -2 * sr * shi
    hr1 = hr = hi1 = hi = 0
    for t in range(degree, 0, -1):
        hr2, hr1, hi2, hi1 = hr1, hr, hi1, hi
        z = ar[t - 1] + r * hr1 - i * hi - hr2
        hi = i * hr1 + r * hi1 - hi2
        hr = z
    r = sr * chi
    i = cr * shi
    return r * hr - i * hi, r * hi + i * hr


def cart2utm(X, Y, Z, zone):
    """geoFunctions/__init__.py:176-372 -> (E, N, U) on the International 1924 ellipsoid (ED50 shift applied)."""
    a, f = 6378388.0, 1.0 / 297.0
    ex2 = (2 - f) * f / (1 - f) ** 2
    c = a * np.sqrt(1 + ex2)
    vec = np.array([X, Y, Z - 4.5])
    alpha = 7.56e-07
    R = np.array([[1, -alpha, 0], [alpha, 1, 0], [0, 0, 1]])
    v = 0.9999988 * R.dot(vec) + np.array([89.5, 93.8, 127.6])
    L = np.arctan2(v[1], v[0])
    N1 = 6395000.0
    B = np.arctan2(v[2] / ((1 - f) ** 2 * N1), np.linalg.norm(v[0:2]) / N1)
    U, oldU, it = 0.1, 0, 0
    while abs(U - oldU) > 0.0001:
        oldU = U
        N1 = c / np.sqrt(1 + ex2 * (np.cos(B)) ** 2)
        B = np.arctan2(v[2] / ((1 - f) ** 2 * N1 + U), np.linalg.norm(v[0:2]) / (N1 + U))
        U = np.linalg.norm(v[0:2]) / np.cos(B) - N1
        it += 1
        if it > 100:
            break
    m0 = 0.0004
    n = f / (2 - f)
    m = n ** 2 * (1.0 / 4.0 + n ** 2 / 64)
    Q_n = a + (a * (-n - m0 + m * (1 - m0))) / (1 + n)
    L0 = ((zone - 30) * 6 - 3) * np.pi / 180
    bg = np.array([-0.00337077907, 4.73444769e-06, -8.2991457e-09, 1.5878533e-11])
    gtu = np.array([0.000841275991, 7.67306686e-07, 1.2129123e-09, 2.48508228e-12])
    neg = B < 0
    Bg = np.abs(B)
    Bg = Bg + _clsin(bg, 4, 2 * Bg)
    Lg = L - L0
    cos_bn = np.cos(Bg)
    Np = np.arctan2(np.sin(Bg), np.cos(Lg) * cos_bn) * 2
    Ep = np.arctanh(np.sin(Lg) * cos_bn) * 2
    dN, dE = _clksin(gtu, 4, Np, Ep)
    Np = Np / 2 + dN
    Ep = Ep / 2 + dE
    N = Q_n * Np
    E = Q_n * Ep + 500000.0
    if neg:
        N = -N + 20000000
    return E, N, U


def post_navigate(s, prn_rows, status_rows, abs_rows, ip_rows, nav_sol_period=500.0, elevation_mask=10.0,
                  use_trop=True):
    """postNavigation.py:75-305 on tracking output given as rows (PRN, status, absoluteSample series, I_P series):
    -> dict with the reference's navSolutions fields (64 measurement columns) or None where the reference gives up."""
    nch = s.numberOfChannels
    if s.msToProcess < 36000 or sum(1 for x in status_rows if x != '-') < 4:
        return None
    first, active = find_preambles(np.stack(ip_rows), list(status_rows), nch)
    table = np.zeros((32, len(EPH_NAMES)))
    tow = None
    for ch in active:
        bits = [str(int(b)) for b in nav_bits(ip_rows[ch], int(first[ch]))]
        dec, tow = ephemeris(bits[1:], bits[0])
        table[int(prn_rows[ch]) - 1] = [float(v) for v in dec]
    if active.size < 4:
        return None
    out = dict(PRN=np.zeros((nch, 64)), DOP=np.zeros((5, 64)), utmZone=0, firstSubFrame=first, eph=table, TOW=tow)
    for k in ('el', 'az', 'rawP', 'correctedP'):
        out[k] = np.nan * np.ones((nch, 64))
    for k in ('X', 'Y', 'Z', 'dt', 'latitude', 'longitude', 'height', 'E', 'N', 'U'):
        out[k] = np.nan * np.ones(64)
    sat_elev = np.inf * np.ones(nch)
    ready = active.copy()
    t_tx = tow
    prn_arr = np.asarray(prn_rows)
    for m in range(int(np.fix(s.msToProcess - first.max()) / nav_sol_period)):
        act = np.intersect1d((sat_elev >= elevation_mask).nonzero()[0], ready)
        out['PRN'][act, m] = prn_arr[act]
        out['rawP'][:, m] = calculate_pseudoranges(s, abs_rows, first + nav_sol_period * m, act)
        sat, clk = satpos(t_tx, prn_arr[act], table)
        if act.size > 3:
            p, out['el'][act, m], out['az'][act, m], out['DOP'][:, m] = least_square_pos(
                sat, out['rawP'][act, m] + clk * s.c, s.c, use_trop)
            p = np.asarray(p).reshape(-1)
            out['X'][m], out['Y'][m], out['Z'][m], out['dt'][m] = p
            sat_elev = out['el'][:, m]
            out['correctedP'][act, m] = out['rawP'][act, m] + clk * s.c + out['dt'][m]
            out['latitude'][m], out['longitude'][m], out['height'][m] = cart2geo(p[0], p[1], p[2], 4)
            out['utmZone'] = find_utm_zone(out['latitude'][m], out['longitude'][m])
            out['E'][m], out['N'][m], out['U'][m] = cart2utm(p[0], p[1], p[2], out['utmZone'])
        else:
            out['DOP'][:, m] = 0.0
            out['az'][act, m] = np.nan
            out['el'][act, m] = np.nan
        t_tx += nav_sol_period / 1000
    return out
