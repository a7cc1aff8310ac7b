"""ctypes binding of libsgx.so (include/sgx.h).  No torch, no fallback.

The product path fails loudly: if the library is missing, or a device entry point is called
without a GPU, an exception is raised - there is no CPU implementation behind these calls.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGX_LIB") or os.path.join(HERE, "lib", "libsgx.so")   # SGX_LIB: another build of the same library (kernel variants side by side)

SGX_OK = 0
SGX_E_ARG, SGX_E_HIP, SGX_E_NOMEM, SGX_E_INDEX, SGX_E_RCCL, SGX_E_RANGE, SGX_E_DEFER = -1, -2, -3, -4, -5, -6, -7
NUM_SERIES = 13
DT_INT8, DT_INT16, DT_UINT8, DT_FLOAT32 = 0, 1, 2, 3   # sgx_track_ex data_type (include/sgx.h)
DT_FLOAT64, DT_UINT16, DT_INT32, DT_UINT32, DT_INT64, DT_UINT64, DT_FLOAT16 = 4, 5, 6, 7, 8, 9, 10
MAX_SATS = 16
SERIES = ("absoluteSample", "codeFreq", "carrFreq", "I_P", "I_E", "I_L", "Q_E", "Q_P", "Q_L",
          "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt")


class SgxError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libsgx error %d: %s" % (code, msg))
        self.code = code


class Settings(C.Structure):
    _fields_ = [("samplingFreq", C.c_double), ("IF", C.c_double), ("codeFreqBasis", C.c_double),
                ("acqSearchBand", C.c_double), ("acqThreshold", C.c_double),
                ("dllDampingRatio", C.c_double), ("dllNoiseBandwidth", C.c_double),
                ("dllCorrelatorSpacing", C.c_double), ("pllDampingRatio", C.c_double),
                ("pllNoiseBandwidth", C.c_double), ("skipNumberOfBytes", C.c_int64),
                ("codeLength", C.c_int32), ("numberOfChannels", C.c_int32)]


class ChanInit(C.Structure):
    _fields_ = [("acquiredFreq", C.c_double), ("codePhase", C.c_double), ("prn", C.c_int32),
                ("reserved", C.c_int32)]


class Sat(C.Structure):
    _fields_ = [("code_fcw", C.c_uint64), ("code_c0", C.c_uint64), ("nav_seed", C.c_uint64),
                ("car_fcw", C.c_uint32), ("car_ph0", C.c_uint32), ("prn", C.c_int32), ("amp", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_sats", C.c_int32), ("nav_mode", C.c_int32),
                ("sats", Sat * MAX_SATS), ("cos_lut", C.c_int16 * 256), ("nav_bits", (C.c_uint8 * 256) * MAX_SATS)]


class Timing(C.Structure):
    _fields_ = [("acquire_ms", C.c_float), ("acq_coarse_ms", C.c_float), ("acq_fine_ms", C.c_float),
                ("track_ms", C.c_float), ("synth_ms", C.c_float), ("track_kernel", C.c_float),
                ("track_members", C.c_float), ("track_streamed", C.c_float)]


# every symbol include/sgx.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PROTOS = {
    "sgx_version": (C.c_char_p, []),
    "sgx_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "sgx_samples_per_code": (C.c_int, [C.POINTER(Settings), C.POINTER(C.c_int64)]),
    "sgx_generate_ca_code": (C.c_int, [C.c_int32, _P]),
    "sgx_make_ca_table": (C.c_int, [C.POINTER(Settings), _P]),
    "sgx_calc_loop_coef": (C.c_int, [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    "sgx_trk_math_eval": (C.c_int, [C.c_int32, C.c_double, C.c_double, _P]),
    "sgx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sgx_ctx_create": (C.c_int, [C.POINTER(Settings), C.c_int, C.POINTER(_P)]),
    "sgx_ctx_create_prio": (C.c_int, [C.POINTER(Settings), C.c_int, C.c_int, C.POINTER(_P)]),
    "sgx_ctx_destroy": (C.c_int, [_P]),
    "sgx_ctx_sync": (C.c_int, [_P]),
    "sgx_get_timing": (C.c_int, [_P, C.POINTER(Timing)]),
    "sgx_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "sgx_host_free": (C.c_int, [_P]),
    "sgx_if_upload": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(_P)]),
    "sgx_if_upload_file": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_size_t, C.POINTER(_P)]),
    "sgx_if_open_file": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_size_t, C.POINTER(_P)]),
    "sgx_if_wait": (C.c_int, [_P, _P, C.c_size_t]),
    "sgx_if_synth": (C.c_int, [_P, C.POINTER(Scene), C.c_uint64, C.c_size_t, C.POINTER(_P)]),
    "sgx_if_download": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P]),
    "sgx_if_length": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "sgx_if_free": (C.c_int, [_P, _P]),
    "sgx_acquire": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_int32, C.c_int32, C.c_int32,
                              _P, _P, _P, _P, _P]),
    "sgx_acquire_f64": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "sgx_acquire_begin": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_int32, C.c_int32, C.c_int32]),
    "sgx_acquire_end": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "sgx_track_chained": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P,
                                    C.POINTER(C.c_int32)]),
    "sgx_track": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P]),
    "sgx_track_ex": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32]),
    "sgx_track_plan": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sgx_acquire_plan": (C.c_int, [C.c_int32] * 6 + [C.POINTER(C.c_int32)] * 4),
    "sgx_acquire_plan_limits": (C.c_int, [C.POINTER(C.c_int32)] * 2),
    "sgx_stream_rates": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sgx_probe_stats": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_double, _P, _P, _P, C.POINTER(C.c_int32)]),
    "sgx_find_preambles": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sgx_nav_parity_check": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "sgx_check_t": (C.c_int, [C.c_double, _P]),
    "sgx_e_r_corr": (C.c_int, [C.c_double, _P, _P]),
    "sgx_togeod": (C.c_int, [C.c_double] * 5 + [_P, _P, _P]),
    "sgx_topocent": (C.c_int, [_P] * 5),
    "sgx_tropo": (C.c_int, [C.c_double] * 8 + [_P]),
    "sgx_satpos": (C.c_int, [C.c_double, _P, C.c_int32, _P, _P, _P]),
    "sgx_least_square_pos": (C.c_int, [_P, _P, C.c_int32, C.c_double, C.c_int32, _P, _P, _P, _P, _P]),
    "sgx_cart2geo": (C.c_int, [C.c_double] * 3 + [C.c_int32, _P, _P, _P]),
    "sgx_find_utm_zone": (C.c_int, [C.c_double, C.c_double, _P]),
    "sgx_cart2utm": (C.c_int, [C.c_double] * 3 + [C.c_int32, _P, _P, _P]),
    "sgx_ephemeris": (C.c_int, [_P, C.c_int32, C.c_uint8, _P, C.POINTER(C.c_int64)]),
    "sgx_pseudoranges": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_double,
                                   C.c_double, _P]),
    "sgx_nav_bits": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_int32)]),
    "sgx_post_navigate": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int64, C.c_int64,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                    _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int32), _P]),
    "sgx_comm_unique_id": (C.c_int, [_P]),
    "sgx_comm_create": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    "sgx_comm_allgather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "sgx_comm_destroy": (C.c_int, [_P]),
    "sgx_acquire_sharded": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32,
                                      C.c_int32, _P, _P, _P, _P, _P]),
}
SYMBOLS = tuple(sorted(_PROTOS))

_lib = None


def lib():
    """Load libsgx.so (built in-tree by build.py / __graft_entry__.build). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsgx.so is not built (%s missing): run `python __graft_entry__.py` or "
                              "`python softgnss-python_amd/build.py`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            f = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().sgx_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(code):
    if code != SGX_OK:
        msg = last_error()
        if code == SGX_E_INDEX:
            raise IndexError(msg)          # the reference raises IndexError here (acquisition.py:152-162)
        raise SgxError(code, msg)


def settings_struct(s):
    """POD mirror of a Settings-like object (attribute names of reference initialize.py:85-173)."""
    return Settings(float(s.samplingFreq), float(s.IF), float(s.codeFreqBasis), float(s.acqSearchBand),
                    float(s.acqThreshold), float(s.dllDampingRatio), float(s.dllNoiseBandwidth),
                    float(s.dllCorrelatorSpacing), float(s.pllDampingRatio), float(s.pllNoiseBandwidth),
                    int(s.skipNumberOfBytes), int(s.codeLength), int(s.numberOfChannels))


def track_plan(settings, data_type=0, n_channels=8, n_cus=256, float_in_range=False):
    """(kernel, members per channel) sgx_track_ex would run for these settings, sample type and channel count on a device
    with n_cus compute units - the host's one selection rule (csrc/sgx_trk.hip); needs no GPU."""
    st = settings_struct(settings)
    k, mbr = C.c_int32(0), C.c_int32(0)
    check(lib().sgx_track_plan(C.byref(st), int(data_type), int(n_channels), int(n_cus), 1 if float_in_range else 0,
                               C.byref(k), C.byref(mbr)))
    return k.value, mbr.value


def acquire_plan(n_prn=32, n_bins=29, n_blocks=2, noncoh=False, chunk_rows=0, max_queues=2):
    """(PRNs per chunk, runs of Doppler bins per PRN, bins per run, queues): how sgx_acquire cuts the correlation batch -
    the host's one rule (csrc/sgx_acq.hip: acq_plan); needs no GPU."""
    v = [C.c_int32(0) for _ in range(4)]
    check(lib().sgx_acquire_plan(int(n_prn), int(n_bins), int(n_blocks), 1 if noncoh else 0, int(chunk_rows), int(max_queues),
                                 *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def acquire_plan_limits():
    """(default chunk rows, largest batch of rows per launch) of csrc/sgx_acq.hip's chunk rule."""
    a, b = C.c_int32(0), C.c_int32(0)
    check(lib().sgx_acquire_plan_limits(C.byref(a), C.byref(b)))
    return a.value, b.value


def scene_struct(scene):
    sc = Scene()
    sc.seed = scene.seed
    sc.n_sats = len(scene.sats)
    for i, s in enumerate(scene.sats):
        sc.sats[i] = Sat(s["code_fcw"], s["code_c0"], s["nav_seed"], s["car_fcw"], s["car_ph0"], s["prn"],
                         s["amp"])
    for i in range(256):
        sc.cos_lut[i] = int(scene.cos_lut[i])
    tab = getattr(scene, "nav_bits", None)
    sc.nav_mode = 0 if tab is None else 1
    if tab is not None:
        for i in range(len(scene.sats)):
            packed = np.packbits(np.asarray(tab[i], dtype=np.uint8), bitorder="little")
            for j in range(256):
                sc.nav_bits[i][j] = int(packed[j])
    return sc


class _Pinned(object):
    """Owner of one pinned host allocation.  `busy` is set while a numpy array handed out by pinned_empty() is alive
    (cleared by a weakref finalizer on the ctypes buffer the array is built on)."""

    def __init__(self, nbytes):
        self.ptr = _P()
        check(lib().sgx_host_alloc(int(nbytes), C.byref(self.ptr)))
        self.nbytes = int(nbytes)
        self.busy = False

    def __del__(self):
        try:
            if self.ptr:
                lib().sgx_host_free(self.ptr)
        except Exception:
            pass


_pinned_pool = []   # pinning is slow (ms per 30 MB): allocations whose arrays have died are reused
_pinned_lock = threading.RLock()   # re-entrant: a finalizer may run (cyclic GC) while pinned_empty() holds it


def _pinned_release(own):
    with _pinned_lock:
        own.busy = False


def pinned_empty(shape, dtype=np.float64):
    """numpy array in pinned host memory (falls back to a pageable array if pinning fails)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    own = None
    with _pinned_lock:
        for cand in _pinned_pool:
            if cand.nbytes >= n and not cand.busy:
                own = cand
                break
        if own is not None:
            own.busy = True
    if own is None:
        try:
            own = _Pinned(max(n, 1))
        except Exception:
            return np.empty(shape, dtype=dtype)
        own.busy = True
        with _pinned_lock:
            _pinned_pool.append(own)
            while len(_pinned_pool) > 8:
                # drop an idle allocation (a busy one stays alive through its array even when it leaves the pool)
                idle = [c for c in _pinned_pool if not c.busy and c is not own]
                _pinned_pool.remove(idle[0] if idle else _pinned_pool[0])
    buf = (C.c_char * own.nbytes).from_address(own.ptr.value)
    buf._owner = own                       # ctypes object keeps the owner, numpy keeps the ctypes object
    weakref.finalize(buf, _pinned_release, own)   # every view of the array holds `buf` through .base
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def device_count():
    n = C.c_int(0)
    rc = lib().sgx_device_count(C.byref(n))
    return n.value if rc == SGX_OK else 0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def nav_bits(i_p_row, sub_frame_start):
    """uint8 bits (1 = positive 20-ms sum) of I_P[start-20 : start+30000], reference postNavigation.py:125-138."""
    a = np.ascontiguousarray(i_p_row, dtype=np.float64)
    out = np.zeros(1501, dtype=np.uint8)
    nb = C.c_int32(0)
    rc = lib().sgx_nav_bits(_ptr(a), a.shape[0], int(sub_frame_start), _ptr(out), C.byref(nb))
    if rc == SGX_E_RANGE:
        raise ValueError(last_error())
    check(rc)
    return out[:nb.value]


class Context(object):
    """One device context (hipStream + scratch) per GPU."""

    def __init__(self, settings, device=0, priority=0):
        """priority: stream priority class (-1 high, 0 normal, +1 low); contexts meant to run at the same time on
        one GPU take different classes (they then never share a hardware queue)."""
        self._h = _P()
        self._s = settings_struct(settings)
        check(lib().sgx_ctx_create_prio(C.byref(self._s), int(device), int(priority), C.byref(self._h)))
        self.device = int(device)
        self._records = weakref.WeakSet()   # live records of this context: freed (loader threads joined) before it

    def close(self):
        if self._h:
            for rec in list(self._records):
                rec.free()
            lib().sgx_ctx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().sgx_ctx_sync(self._h))

    def timing(self):
        t = Timing()
        check(lib().sgx_get_timing(self._h, C.byref(t)))
        return dict(acquire_ms=t.acquire_ms, acq_coarse_ms=t.acq_coarse_ms, acq_fine_ms=t.acq_fine_ms,
                    track_ms=t.track_ms, synth_ms=t.synth_ms, track_kernel=int(t.track_kernel),
                    track_members=int(t.track_members), track_streamed=int(t.track_streamed))

    def stream_rates(self, nbytes=1 << 30, reps=5):
        """(read GB/s, copy GB/s) measured on this device: the practical HBM roof next to the 8 TB/s datasheet peak."""
        r, w = C.c_double(0), C.c_double(0)
        check(lib().sgx_stream_rates(self._h, int(nbytes), int(reps), C.byref(r), C.byref(w)))
        return r.value, w.value

    # ---- records ----
    def upload(self, samples):
        """int8 samples - or, for a two-byte record, the file's bytes viewed as int8 (upload_bytes)."""
        a = np.ascontiguousarray(samples, dtype=np.int8)
        h = _P()
        check(lib().sgx_if_upload(self._h, _ptr(a), a.size, C.byref(h)))
        return Record(self, h, a.size)

    def upload_bytes(self, data):
        """The raw bytes of a record of any sample type (bytes / any contiguous array), as they lie in the file."""
        return self.upload(np.frombuffer(memoryview(np.ascontiguousarray(data)).cast('B'), dtype=np.int8))

    def upload_file(self, path, file_offset, n):
        """Stream bytes [file_offset, file_offset+n) of a raw int8 record file into HBM."""
        h = _P()
        check(lib().sgx_if_upload_file(self._h, os.fsencode(path), int(file_offset), int(n), C.byref(h)))
        ln = C.c_size_t(0)
        check(lib().sgx_if_length(h, C.byref(ln)))
        return Record(self, h, int(ln.value))

    def open_file(self, path, file_offset, n):
        """Like upload_file, but returns at once: the record fills in the background (file order) while
        acquisition and tracking already run on it.  Record.wait() blocks until everything is resident."""
        h = _P()
        check(lib().sgx_if_open_file(self._h, os.fsencode(path), int(file_offset), int(n), C.byref(h)))
        ln = C.c_size_t(0)
        check(lib().sgx_if_length(h, C.byref(ln)))
        return Record(self, h, int(ln.value))

    def synth(self, scene, n, offset=0):
        h = _P()
        sc = scene_struct(scene)
        check(lib().sgx_if_synth(self._h, C.byref(sc), int(offset), int(n), C.byref(h)))
        return Record(self, h, int(n))

    # ---- hot path ----
    def acquire(self, rec, offset, n_samples, prn0, n_blocks=2, noncoh=False):
        self._acq_token = getattr(self, "_acq_token", 0) + 1   # (any search supersedes a deferred one nobody has looked at)
        prn = np.ascontiguousarray(prn0, dtype=np.int32)
        n = prn.size
        carr = np.zeros(n)
        cph = np.zeros(n)
        met = np.zeros(n)
        fb = np.zeros(n, dtype=np.int32)
        fi = np.zeros(n, dtype=np.int32)
        check(lib().sgx_acquire(self._h, rec._h, int(offset), int(n_samples), _ptr(prn), n, int(n_blocks),
                                1 if noncoh else 0, _ptr(carr), _ptr(cph), _ptr(met), _ptr(fb), _ptr(fi)))
        return dict(carrFreq=carr, codePhase=cph, peakMetric=met, freqBin=fb, fineIdx=fi)

    # ---- the same without host round trips between the stages (include/sgx.h, "round 6") ----
    def acquire_begin(self, rec, offset, n_samples, prn0, n_blocks=2, noncoh=False):
        """Queue the search of acquire(); nothing is looked at.  acquire_end() returns what acquire() would have."""
        prn = np.ascontiguousarray(prn0, dtype=np.int32)
        check(lib().sgx_acquire_begin(self._h, rec._h, int(offset), int(n_samples), _ptr(prn), prn.size, int(n_blocks),
                                      1 if noncoh else 0))
        self._acq_token = getattr(self, "_acq_token", 0) + 1   # (one search may be pending per context)
        return self._acq_token

    def acquire_end(self, n):
        carr = np.zeros(n)
        cph = np.zeros(n)
        met = np.zeros(n)
        fb = np.zeros(n, dtype=np.int32)
        fi = np.zeros(n, dtype=np.int32)
        check(lib().sgx_acquire_end(self._h, _ptr(carr), _ptr(cph), _ptr(met), _ptr(fb), _ptr(fi)))
        return dict(carrFreq=carr, codePhase=cph, peakMetric=met, freqBin=fb, fineIdx=fi)

    def track_chained(self, rec, n_ch, ms, rec_file_offset=0, data_type=DT_INT8):
        """preRun on the device behind the pending acquisition + the tracking kernel behind it, one wait.
        Returns None where the queued sequence does not apply (SGX_E_DEFER: run acquire_end, preRun and track instead),
        else (series[n_ch, 13, ms], ms_done[n_ch], PRN[n_ch], acquiredFreq[n_ch], codePhase[n_ch], n_active)."""
        out = pinned_empty((int(n_ch), NUM_SERIES, int(ms)))
        done = np.zeros(n_ch, dtype=np.int32)
        prn = np.zeros(n_ch, dtype=np.int32)
        freq = np.zeros(n_ch)
        cph = np.zeros(n_ch)
        n_act = C.c_int32(0)
        rc = lib().sgx_track_chained(self._h, rec._h, int(rec_file_offset), int(n_ch), int(ms), _ptr(out), _ptr(done),
                                     int(data_type), _ptr(prn), _ptr(freq), _ptr(cph), C.byref(n_act))
        if rc == SGX_E_DEFER:
            return None
        check(rc)
        return out, done, prn, freq, cph, n_act.value

    def acquire_sharded(self, comm, rank, world, rec, offset, n_samples, n_prn_total=32, n_blocks=2, noncoh=False):
        """This rank's share of the PRN search + the peak gather as ONE library call (sgx_acquire_sharded): packed on the
        device, one ncclAllGather (comm: a Comm, or None for no collective), one look.  Returns the merged 32-entry arrays."""
        self._acq_token = getattr(self, "_acq_token", 0) + 1
        carr = np.zeros(32)
        cph = np.zeros(32)
        met = np.zeros(32)
        fb = np.zeros(32, dtype=np.int32)
        fi = np.zeros(32, dtype=np.int32)
        check(lib().sgx_acquire_sharded(self._h, comm._h if comm is not None else None, int(rank), int(world), rec._h,
                                        int(offset), int(n_samples), int(n_prn_total), int(n_blocks), 1 if noncoh else 0,
                                        _ptr(carr), _ptr(cph), _ptr(met), _ptr(fb), _ptr(fi)))
        return dict(carrFreq=carr, codePhase=cph, peakMetric=met, freqBin=fb.astype(np.int64), fineIdx=fi.astype(np.int64))

    def acquire_f64(self, signal, prn0, n_blocks=2, noncoh=False):
        """acquire() on a host signal of any real dtype (copied to HBM as fp64)."""
        self._acq_token = getattr(self, "_acq_token", 0) + 1
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        prn = np.ascontiguousarray(prn0, dtype=np.int32)
        n = prn.size
        carr = np.zeros(n)
        cph = np.zeros(n)
        met = np.zeros(n)
        fb = np.zeros(n, dtype=np.int32)
        fi = np.zeros(n, dtype=np.int32)
        check(lib().sgx_acquire_f64(self._h, _ptr(sig), sig.size, _ptr(prn), n, int(n_blocks), 1 if noncoh else 0,
                                    _ptr(carr), _ptr(cph), _ptr(met), _ptr(fb), _ptr(fi)))
        return dict(carrFreq=carr, codePhase=cph, peakMetric=met, freqBin=fb, fineIdx=fi)

    def probe_stats(self, rec, offset, n, fs_mhz):
        """(f, Pxx, hist, n_segments) of the record window: Welch PSD and histogram of Settings.probeData."""
        f = np.zeros(8193)
        pxx = np.zeros(8193)
        hist = np.zeros(255, dtype=np.int64)
        nseg = C.c_int32(0)
        rc = lib().sgx_probe_stats(self._h, rec._h, int(offset), int(n), float(fs_mhz), _ptr(f), _ptr(pxx),
                                   _ptr(hist), C.byref(nseg))
        if rc == SGX_E_RANGE:
            raise ValueError(last_error())
        check(rc)
        return f, pxx, hist, nseg.value

    def find_preambles(self, i_p, search_start=0):
        """i_p: float64[n_ch, ms] -> int array firstSubFrame[n_ch] (0 = no verified preamble)."""
        a = np.ascontiguousarray(i_p, dtype=np.float64)
        out = np.zeros(a.shape[0], dtype=np.int32)
        rc = lib().sgx_find_preambles(self._h, _ptr(a), a.shape[0], a.shape[1], int(search_start), _ptr(out))
        if rc == SGX_E_RANGE:      # the exceptions the reference's numpy code raises on a record cut short
            msg = last_error()
            raise (IndexError if msg.startswith("IndexError") else ValueError)(msg)
        check(rc)
        return out.astype(int)

    def track(self, rec, chans, ms, rec_file_offset=0, data_type=DT_INT8):
        """chans: sequence of (prn, acquiredFreq, codePhase). Returns (series[n_ch,13,ms], ms_done).
        data_type DT_INT16: `rec` holds the BYTES of a little-endian int16 file (see sgx_track_ex in include/sgx.h)."""
        n = len(chans)
        arr = (ChanInit * n)()
        for i, (prn, f, cp) in enumerate(chans):
            arr[i] = ChanInit(float(f), float(cp), int(prn), 0)
        out = pinned_empty((n, NUM_SERIES, int(ms)))
        done = np.zeros(n, dtype=np.int32)
        check(lib().sgx_track_ex(self._h, rec._h, int(rec_file_offset), C.cast(arr, _P), n, int(ms), _ptr(out),
                                 _ptr(done), int(data_type)))
        return out, done


class Record(object):
    """int8 IF record resident in HBM."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = n
        ctx._records.add(self)

    def __len__(self):
        return self.n

    def wait(self, n=0):
        """Block until the first n samples (0 = all) of a record opened with Context.open_file are resident."""
        check(lib().sgx_if_wait(self.ctx._h, self._h, int(n)))

    def download(self, offset=0, n=None):
        n = self.n - offset if n is None else n
        out = np.empty(n, dtype=np.int8)
        check(lib().sgx_if_download(self.ctx._h, self._h, int(offset), int(n), _ptr(out)))
        return out

    def free(self):
        if self._h:
            # (a context that is already gone: sgx_if_free(NULL, h) still joins the loader thread and frees the HBM)
            lib().sgx_if_free(self.ctx._h if self.ctx._h else None, self._h)
        self._h = _P()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm(object):
    """RCCL communicator for the acquisition peak gather (one process per GPU)."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        self._h = _P()
        self.n_ranks = n_ranks
        uid = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        check(lib().sgx_comm_create(ctx._h, n_ranks, rank, C.cast(uid, _P), C.byref(self._h)))

    @staticmethod
    def unique_id():
        uid = (C.c_uint8 * 128)()
        check(lib().sgx_comm_unique_id(C.cast(uid, _P)))
        return bytes(uid)

    def allgather(self, payload):
        send = np.ascontiguousarray(payload).view(np.uint8).ravel()
        recv = np.empty(send.size * self.n_ranks, dtype=np.uint8)
        check(lib().sgx_comm_allgather(self._h, _ptr(send), _ptr(recv), send.size))
        return recv.reshape(self.n_ranks, send.size)

    def close(self):
        if self._h:
            lib().sgx_comm_destroy(self._h)
            self._h = _P()
