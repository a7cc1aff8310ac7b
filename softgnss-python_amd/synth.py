"""Deterministic, integer-only synthetic GPS L1 C/A IF record generator (host side).

The reference ships no sample data (its default input is a path on the author's
laptop, reference initialize.py:99), so parity tests and the bench need their own
record.  Every sample is a pure function of (scene, sample index) computed with
integer arithmetic only, so this numpy generator and the HIP generator in
csrc/sgx_synth.hip produce bit-identical int8 streams.

sample(n) = clip( noise(n) + sum_sat round(A * ca[chip(n)] * nav(n) * cosLUT[phase32(n) >> 24] / 128), -127, 127 )

  noise(n)   : Irwin-Hall sum of 4 bytes of splitmix64(seed + (n+1)*GOLDEN), centred, * 35 >> 8  (sigma ~ 20 LSB)
  phase32(n) : 32-bit carrier NCO, (ph0 + n * car_fcw) mod 2^32
  chip(n)    : 32.32 fixed-point code NCO, ((n * code_fcw + code_c0) >> 32) mod 1023
  nav(n)     : +-1 from splitmix64(nav_seed + (bit+1)*GOLDEN) & 1, one bit per 20 code periods
               (hash mode), or 2*table[bit mod 2048]-1 (structured mode: 300-bit subframes with the TLM
               preamble and IS-GPS-200 parity, so that the reference's findPreambles has something to find)
"""
import numpy as np

GOLDEN = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1
MAX_SATS = 16
NAV_TABLE_BITS = 2048     # per-satellite navigation bit table (structured mode), repeats after 40.96 s
NOISE_MUL = 35
NOISE_SHIFT = 8
L1_HZ = 1575.42e6


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wraps mod 2^64)."""
    x = np.asarray(x, dtype=np.uint64)
    z = x.copy()
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def ca_code_bits(prn0):
    """1023-chip C/A Gold code as +-1 int8, PRN index 0..31 (IS-GPS-200 G2 delay form).

    Integer LFSR statement of the sequence the reference builds in
    initialize.py:234-302 (G1 taps 3,10; G2 taps 2,3,6,8,9,10; G2 delayed by g2s).
    """
    g2s = [5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
           469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862]
    if not 0 <= prn0 < 32:
        raise ValueError("prn index must be in 0..31")
    r1 = [1] * 10
    r2 = [1] * 10
    g1 = np.empty(1023, dtype=np.int8)
    g2 = np.empty(1023, dtype=np.int8)
    for i in range(1023):
        g1[i] = r1[9]
        g2[i] = r2[9]
        f1 = r1[2] ^ r1[9]
        f2 = r2[1] ^ r2[2] ^ r2[5] ^ r2[7] ^ r2[8] ^ r2[9]
        r1 = [f1] + r1[:9]
        r2 = [f2] + r2[:9]
    g2 = np.roll(g2, g2s[prn0])
    # bit 1 <-> chip +1 (ICD first-10-chips octal convention)
    return (1 - 2 * (g1 ^ g2 ^ 1)).astype(np.int8)


def gps_parity(d, d29s, d30s):
    """IS-GPS-200 Table 20-XIV: 30 transmitted bits of a word from its 24 data bits d[0..23] and the last two
    transmitted bits of the previous word."""
    D = [b ^ d30s for b in d]

    def x(*idx):
        v = 0
        for i in idx:
            v ^= d[i - 1]
        return v

    D.append(d29s ^ x(1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23))
    D.append(d30s ^ x(2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24))
    D.append(d29s ^ x(1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22))
    D.append(d30s ^ x(2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23))
    D.append(d30s ^ x(1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24))
    D.append(d29s ^ x(3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24))
    return D


def subframe_bits(seed, first_boundary=100, n_bits=NAV_TABLE_BITS):
    """Bit table whose subframes (10 words x 30 bits, word 1 starting with the preamble 10001011) start at
    table positions first_boundary + 300 k; every word carries valid parity."""
    import random
    rng = random.Random(int(seed))
    out = []
    d29s = d30s = 0
    n_sub = (n_bits + 300) // 300 + 2
    for _ in range(n_sub):
        for wno in range(10):
            d = [rng.randint(0, 1) for _ in range(24)]
            if wno == 0:
                d[:8] = [1, 0, 0, 0, 1, 0, 1, 1]
            w = gps_parity(d, d29s, d30s)
            d29s, d30s = w[28], w[29]
            out.extend(w)
    skip = 300 - (first_boundary % 300)     # table bit 0 is `skip` bits into a subframe
    return np.array(out[skip:skip + n_bits], dtype=np.uint8)


GPS_PI = 3.1415926535898            # the value the navigation message is scaled with (IS-GPS-200)

# (name, scale exponent, times pi, signed, bit slices in the 300-bit subframe) per subframe, laid out where the
# reference's decoder reads them (reference ephemeris.py:106-175, which departs from IS-GPS-200 for IODC / T_GD)
EPH_LAYOUT = {
    1: [("weekNumber", 0, 0, 0, [(60, 70)]), ("accuracy", 0, 0, 0, [(72, 76)]), ("health", 0, 0, 0, [(76, 82)]),
        ("T_GD", -31, 0, 1, [(195, 204)]), ("t_oc", 4, 0, 0, [(218, 234)]), ("a_f2", -55, 0, 1, [(240, 248)]),
        ("a_f1", -43, 0, 1, [(248, 264)]), ("a_f0", -31, 0, 1, [(270, 292)])],
    2: [("IODE_sf2", 0, 0, 0, [(60, 68)]), ("C_rs", -5, 0, 1, [(68, 84)]), ("deltan", -43, 1, 1, [(90, 106)]),
        ("M_0", -31, 1, 1, [(106, 114), (120, 144)]), ("C_uc", -29, 0, 1, [(150, 166)]),
        ("e", -33, 0, 0, [(166, 174), (180, 204)]), ("C_us", -29, 0, 1, [(210, 226)]),
        ("sqrtA", -19, 0, 0, [(226, 234), (240, 264)]), ("t_oe", 4, 0, 0, [(270, 286)])],
    3: [("C_ic", -29, 0, 1, [(60, 76)]), ("omega_0", -31, 1, 1, [(76, 84), (90, 114)]), ("C_is", -29, 0, 1, [(120, 136)]),
        ("i_0", -31, 1, 1, [(136, 144), (150, 174)]), ("C_rc", -5, 0, 1, [(180, 196)]),
        ("omega", -31, 1, 1, [(196, 204), (210, 234)]), ("omegaDot", -43, 1, 1, [(240, 264)]),
        ("IODE_sf3", 0, 0, 0, [(270, 278)]), ("iDot", -43, 1, 1, [(278, 292)])],
}


def make_ephemeris(seed, toe=100800, week=1900):
    """Plausible GPS orbit / clock parameters (dict, reference field names) from a seed: near-circular 26 560 km
    orbit at 55 degrees inclination, random node / perigee / anomaly."""
    import random
    rng = random.Random(int(seed) ^ 0x657068)
    pi = GPS_PI
    return dict(weekNumber=week, accuracy=1, health=0, T_GD=rng.uniform(-1.2e-8, 1.2e-8), t_oc=toe,
                a_f2=0.0, a_f1=rng.uniform(-1e-11, 1e-11), a_f0=rng.uniform(-4e-4, 4e-4),
                IODE_sf2=rng.randrange(256), C_rs=rng.uniform(-120, 120), deltan=rng.uniform(4.0e-9, 5.2e-9),
                M_0=rng.uniform(-pi, pi), C_uc=rng.uniform(-6e-6, 6e-6), e=rng.uniform(0.002, 0.02),
                C_us=rng.uniform(-6e-6, 9e-6), sqrtA=5153.65 + rng.uniform(-0.15, 0.15), t_oe=toe,
                C_ic=rng.uniform(-2e-7, 2e-7), omega_0=rng.uniform(-pi, pi), C_is=rng.uniform(-2e-7, 2e-7),
                i_0=rng.uniform(0.94, 0.985), C_rc=rng.uniform(180, 330), omega=rng.uniform(-pi, pi),
                omegaDot=rng.uniform(-8.6e-9, -7.6e-9), IODE_sf3=0, iDot=rng.uniform(-6e-10, 6e-10))


def encode_ephemeris(eph):
    """{1: uint8[300], 2: ..., 3: ...} with the parameter bits set (everything else 0, parity not filled in) so that
    the reference's decoder returns them, quantised to the message LSBs.  weekNumber is sent modulo 1024."""
    out = {}
    for sid, fields in EPH_LAYOUT.items():
        sub = np.zeros(300, dtype=np.uint8)
        for name, exp, times_pi, signed, slices in fields:
            v = eph["IODE_sf2"] if name == "IODE_sf3" else eph[name]
            if name == "weekNumber":
                v = v - 1024
            width = sum(b - a for a, b in slices)
            q = int(round(v / (GPS_PI if times_pi else 1.0) / 2.0 ** exp))
            lo, hi = (-(1 << (width - 1)), (1 << (width - 1)) - 1) if signed else (0, (1 << width) - 1)
            if not lo <= q <= hi:
                raise ValueError("%s = %r does not fit its %d-bit field" % (name, v, width))
            q &= (1 << width) - 1
            pos = [p for a, b in slices for p in range(a, b)]
            for k, p in enumerate(pos):
                sub[p] = (q >> (width - 1 - k)) & 1
        out[sid] = sub
    return out


def nav_message_bits(seed, first_boundary=100, n_bits=NAV_TABLE_BITS, tow0=1000, first_id=1, eph=None):
    """Like subframe_bits, with a decodable frame structure: word 1 = TLM (preamble + 16 message bits), word 2 =
    HOW (17-bit TOW count of the NEXT subframe, 2 flag bits, 3-bit subframe ID cycling 1..5, 2 filler bits), words
    3-10 random data - so subframes 1, 2 and 3 parse into (random but well-defined) clock and orbit fields - or,
    with `eph` (dict of the reference's field names), the encoded parameters in subframes 1-3.  The subframe that
    starts at table position first_boundary has ID first_id and announces TOW count tow0 + 1."""
    import random
    rng = random.Random(int(seed) ^ 0x6E6176)
    out = []
    d29s = d30s = 0
    n_sub = (n_bits + 300) // 300 + 2
    fixed = encode_ephemeris(eph) if eph is not None else {}
    # the stream is cut so that table bit first_boundary starts a subframe; k counts subframes from that one
    k0 = -((first_boundary + 299) // 300)              # index of the first generated subframe relative to it
    for k in range(k0, k0 + n_sub):
        sid = (first_id - 1 + k) % 5 + 1
        tow = (tow0 + 1 + k) & 0x1FFFF
        for wno in range(10):
            d = [rng.randint(0, 1) for _ in range(24)]
            if wno == 0:
                d[:8] = [1, 0, 0, 0, 1, 0, 1, 1]
            elif wno == 1:
                d[:17] = [(tow >> (16 - b)) & 1 for b in range(17)]
                d[19:22] = [(sid >> 2) & 1, (sid >> 1) & 1, sid & 1]
            elif sid in fixed:
                d = [int(b) for b in fixed[sid][30 * wno:30 * wno + 24]]
            w = gps_parity(d, d29s, d30s)
            d29s, d30s = w[28], w[29]
            out.extend(w)
    skip = (-k0) * 300 - first_boundary                 # generated bits before table bit 0
    return np.array(out[skip:skip + n_bits], dtype=np.uint8)


class Scene(object):
    """Integer description of a synthetic record. All fields are plain Python ints."""

    def __init__(self, seed, sats, fs, cos_lut=None, nav_bits=None):
        self.seed = int(seed) & MASK64
        self.fs = float(fs)
        self.sats = sats  # list of dicts: prn, amp, car_fcw, car_ph0, code_fcw, code_c0, nav_seed
        if cos_lut is None:
            k = np.arange(256)
            cos_lut = np.round(127.0 * np.cos(2.0 * np.pi * (k + 0.5) / 256.0)).astype(np.int16)
        self.cos_lut = np.asarray(cos_lut, dtype=np.int16)
        assert len(self.sats) <= MAX_SATS
        # structured navigation data: uint8[n_sats, NAV_TABLE_BITS] of 0/1, or None for hash bits
        self.nav_bits = None if nav_bits is None else np.asarray(nav_bits, dtype=np.uint8).reshape(len(sats), NAV_TABLE_BITS)

    def with_subframes(self, first_boundary=100):
        """Same scene with structured navigation data (subframes + parity) instead of hash bits."""
        tab = np.stack([subframe_bits(self.seed * 131 + s["prn"], first_boundary) for s in self.sats])
        return Scene(self.seed, self.sats, self.fs, self.cos_lut, tab)

    def with_nav_message(self, first_boundary=100, tow0=1000, first_id=1, ephs=None):
        """Same scene with decodable navigation frames (TLM/HOW with subframe IDs and TOW counts, valid parity);
        ephs: optional {prn: parameter dict} to transmit in subframes 1-3."""
        tab = np.stack([nav_message_bits(self.seed * 137 + s["prn"], first_boundary, NAV_TABLE_BITS, tow0, first_id,
                                         None if ephs is None else ephs.get(s["prn"]))
                        for s in self.sats])
        return Scene(self.seed, self.sats, self.fs, self.cos_lut, tab)

    @staticmethod
    def make(seed, fs, IF, prns, dopplers, code_starts, amps, fc=1.023e6):
        """Build a scene from physical parameters (exact rational arithmetic on the host)."""
        from fractions import Fraction
        sats = []
        for prn, fd, start, amp in zip(prns, dopplers, code_starts, amps):
            car = Fraction(IF) + Fraction(fd)
            car_fcw = int(round(car / Fraction(fs) * (1 << 32))) & 0xFFFFFFFF
            chip_rate = Fraction(fc) * (1 + Fraction(fd) / Fraction(L1_HZ))
            code_fcw = int(round(chip_rate / Fraction(fs) * (1 << 32)))
            period = 1023 << 32
            code_c0 = (-int(start) * code_fcw) % period
            h = int(splitmix64(np.array([(seed + 977 * prn) & MASK64], dtype=np.uint64))[0])
            sats.append(dict(prn=int(prn), amp=int(amp), car_fcw=car_fcw,
                             car_ph0=(h >> 16) & 0xFFFFFFFF, code_fcw=code_fcw, code_c0=code_c0,
                             nav_seed=(h ^ 0xA5A5A5A5DEADBEEF) & MASK64))
        return Scene(seed, sats, fs)

    @staticmethod
    def default(fs=38192000.0, IF=9548000.0, n_sats=8):
        """SURVEY.md section 8(d) default scene: seed 0x5EED0001, 8 satellites."""
        prns = [1, 3, 7, 11, 14, 19, 22, 31][:n_sats]
        dop = [1250, -3100, 4800, -650, 2900, -4400, 350, -1900][:n_sats]
        starts = [12345, 30001, 5, 20000, 777, 38000, 15000, 9000][:n_sats]
        amps = [8, 7, 6, 7, 8, 6, 7, 6][:n_sats]
        return Scene.make(0x5EED0001, fs, IF, prns, dop, starts, amps)


def generate(scene, n, offset=0, chunk=1 << 21):
    """Return int8[n] samples [offset, offset+n) of the scene (numpy, chunked)."""
    out = np.empty(n, dtype=np.int8)
    codes = {s["prn"]: ca_code_bits(s["prn"] - 1).astype(np.int64) for s in scene.sats}
    lut = scene.cos_lut.astype(np.int64)
    g = np.uint64(GOLDEN)
    with np.errstate(over="ignore"):
        for a in range(0, n, chunk):
            m = min(chunk, n - a)
            idx = np.arange(offset + a, offset + a + m, dtype=np.uint64)
            h = splitmix64(np.uint64(scene.seed) + (idx + np.uint64(1)) * g)
            s4 = ((h & np.uint64(0xFF)) + ((h >> np.uint64(8)) & np.uint64(0xFF)) +
                  ((h >> np.uint64(16)) & np.uint64(0xFF)) + ((h >> np.uint64(24)) & np.uint64(0xFF)))
            acc = ((s4.astype(np.int64) - 510) * NOISE_MUL) >> NOISE_SHIFT
            for si, s in enumerate(scene.sats):
                cp = idx * np.uint64(s["code_fcw"]) + np.uint64(s["code_c0"])
                chipw = cp >> np.uint64(32)
                chip = (chipw % np.uint64(1023)).astype(np.int64)
                bit = chipw // np.uint64(1023 * 20)
                if scene.nav_bits is not None:
                    nav = 2 * scene.nav_bits[si][(bit % np.uint64(NAV_TABLE_BITS)).astype(np.int64)].astype(np.int64) - 1
                else:
                    navh = splitmix64(np.uint64(s["nav_seed"]) + (bit + np.uint64(1)) * g)
                    nav = 1 - 2 * (navh & np.uint64(1)).astype(np.int64)
                ph = (np.uint64(s["car_ph0"]) + idx * np.uint64(s["car_fcw"])) & np.uint64(0xFFFFFFFF)
                c = lut[(ph >> np.uint64(24)).astype(np.int64)]
                acc += (s["amp"] * codes[s["prn"]][chip] * nav * c + 64) >> 7
            out[a:a + m] = np.clip(acc, -127, 127).astype(np.int8)
    return out


def record_length(settings_samples_per_code, ms):
    """Bytes needed so that `ms` tracking blocks plus the 11 ms acquisition window always fit."""
    n = int(settings_samples_per_code)
    return (int(ms) + 1) * (n + 1) + n
