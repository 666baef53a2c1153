"""ctypes front-end of the CPU oracle (oracle/hpfw_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (hpfw_amd).  PARITY UNPINNED: see hpfw_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhpfw_oracle.so")

BINS, CTX, LAG, NFILT, FRAME = 121, 20, 80, 64, 2420
MAXRADIX = 24


class PlanInfo(ctypes.Structure):
    _fields_ = [("n_samples", ctypes.c_int64), ("n1", ctypes.c_int64), ("n2", ctypes.c_int64),
                ("h", ctypes.c_int64), ("kmin", ctypes.c_int64), ("kmax", ctypes.c_int64),
                ("m", ctypes.c_int64), ("c", ctypes.c_int64), ("n_frames", ctypes.c_int64),
                ("n_hp", ctypes.c_int64), ("n_radix", ctypes.c_int32),
                ("radix", ctypes.c_int32 * MAXRADIX)]


HIT_DTYPE = np.dtype([("dist", "<u4"), ("clip", "<u4"), ("offset", "<i4"), ("pad", "<u4")])


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("hpfw_oracle.c", "hpfw_oracle.h", "Makefile"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.hpfw_oracle_plan_create.restype = vp
        L.hpfw_oracle_plan_create.argtypes = [i64]
        L.hpfw_oracle_plan_create2.restype = vp
        L.hpfw_oracle_plan_create2.argtypes = [i64, i32]
        L.hpfw_oracle_plan_create3.restype = vp
        L.hpfw_oracle_plan_create3.argtypes = [i64, i32, ctypes.c_uint32]
        L.hpfw_oracle_chirpz_table.restype = i64
        L.hpfw_oracle_chirpz_table.argtypes = [vp, i32, vp]
        L.hpfw_oracle_plan_destroy.argtypes = [vp]
        L.hpfw_oracle_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
        L.hpfw_oracle_plan_bands.argtypes = [vp, vp, vp, vp]
        L.hpfw_oracle_spectrum.argtypes = [vp, vp, vp]
        L.hpfw_oracle_cqmag.argtypes = [vp, vp, vp]
        L.hpfw_oracle_db.argtypes = [vp, i64, vp]
        L.hpfw_oracle_project.argtypes = [vp, vp, i64, vp]
        L.hpfw_oracle_project_q.argtypes = [vp, vp, i64, vp]
        L.hpfw_oracle_pack_q.argtypes = [vp, i64, vp]
        L.hpfw_oracle_delta_q.argtypes = [vp, vp, i64, vp]
        L.hpfw_oracle_quantise_db.argtypes = [vp, i64, vp]
        L.hpfw_oracle_quantise_filters.argtypes = [vp, vp]
        L.hpfw_oracle_set_projection.argtypes = [i32]
        L.hpfw_oracle_get_projection.restype = i32
        L.hpfw_oracle_pack.argtypes = [vp, i64, vp]
        L.hpfw_oracle_project_cfg.argtypes = [vp, vp, i32, i32, i32, i64, i64, vp]
        L.hpfw_oracle_pack_cfg.argtypes = [vp, i32, i32, i64, i64, vp]
        L.hpfw_oracle_extract.restype = i64
        L.hpfw_oracle_extract.argtypes = [vp, vp, vp, vp]
        L.hpfw_oracle_extract_batch.restype = i64
        L.hpfw_oracle_extract_batch.argtypes = [vp, vp, vp, i64, vp, i32]
        L.hpfw_oracle_match_clip.argtypes = [vp, i64, vp, i64, vp, vp]
        L.hpfw_oracle_search_topk.argtypes = [vp, vp, i64, vp, vp, i64, i32, vp, i32]
        L.hpfw_oracle_log10.restype = ctypes.c_double
        L.hpfw_oracle_mel_create.restype = vp
        L.hpfw_oracle_mel_destroy.argtypes = [vp]
        L.hpfw_oracle_mel_tables.argtypes = [vp, vp, vp]
        L.hpfw_oracle_mel_frames.argtypes = [i64]
        L.hpfw_oracle_mel_frames.restype = i64
        L.hpfw_oracle_mel_power.argtypes = [vp, vp, i64, vp, vp]
        L.hpfw_oracle_mel_spectrogram.argtypes = [vp, vp, i64, vp]
        L.hpfw_oracle_mel_spectrogram.restype = i64
        L.hpfw_oracle_knn_windows.argtypes = [vp, vp, i64, vp, i64, i32, i32, vp]
        L.hpfw_oracle_vote_windows.argtypes = [vp, i64, i32, vp, i64, vp]
        L.hpfw_oracle_log10.argtypes = [ctypes.c_double]
        L.hpfw_oracle_twiddle.argtypes = [i64, i64, vp, vp]
        L.hpfw_oracle_fft_dif.argtypes = [vp, i64, vp, i32]
        L.hpfw_oracle_fft_idit.argtypes = [vp, i64, vp, i32]
        L.hpfw_oracle_digit_pos.restype = i64
        L.hpfw_oracle_digit_pos.argtypes = [i64, i64, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Plan:
    """Geometry + tables for clips of n_samples samples (essentia NSGConstantQ, cqt.h:54-61)."""

    def __init__(self, n_samples, force_bluestein=False, conventions=0):
        self._h = lib().hpfw_oracle_plan_create3(int(n_samples), int(bool(force_bluestein)), int(conventions))
        if not self._h:
            raise ValueError(f"unsupported clip length {n_samples} (too short or too long)")
        info = PlanInfo()
        lib().hpfw_oracle_plan_get_info(self._h, ctypes.byref(info))
        for name, _ in PlanInfo._fields_:
            if name != "radix":
                setattr(self, name, int(getattr(info, name)))
        self.radix = [int(info.radix[i]) for i in range(info.n_radix)]
        self.start = np.zeros(BINS, np.int32)
        self.lg = np.zeros(BINS, np.int32)
        self.psize = np.zeros(BINS, np.int32)
        lib().hpfw_oracle_plan_bands(self._h, _p(self.start), _p(self.lg), _p(self.psize))

    def __del__(self):
        if getattr(self, "_h", None):
            try:                                   # (at interpreter exit the module's globals may be gone already)
                lib().hpfw_oracle_plan_destroy(self._h)
            except TypeError:
                pass
            self._h = None

    def chirpz_table(self, which):
        """S15 table of a chirp-z plan as complex64: 0 w, 1 T_L, 2 Bhat ([n1 * n2]), 3 w[k] / L ([kmax - kmin]); 4 (every
        plan): the constant-Q stage's windows G_j, bands concatenated (S5)"""
        count = lib().hpfw_oracle_chirpz_table(self._h, which, None)
        if count == 0:
            raise ValueError("no chirp-z tables: the length takes the mixed-radix transform")
        out = np.zeros(count, np.float32)
        lib().hpfw_oracle_chirpz_table(self._h, which, _p(out))
        return out.view(np.complex64)

    # ---- stages -------------------------------------------------------------------------
    def spectrum(self, pcm):
        pcm = _c(pcm, np.int16)
        assert pcm.shape == (self.n_samples,)
        x = np.zeros((self.kmax - self.kmin, 2), np.float32)
        lib().hpfw_oracle_spectrum(self._h, _p(pcm), _p(x))
        return x

    def cqmag(self, x):
        x = _c(x, np.float32)
        mag = np.zeros((BINS, self.c), np.float32)
        lib().hpfw_oracle_cqmag(self._h, _p(x), _p(mag))
        return mag

    def extract(self, filters, pcm):
        f = _c(filters, np.float32)
        pcm = _c(pcm, np.int16)
        hp = np.zeros(max(self.n_hp, 0), np.uint64)
        lib().hpfw_oracle_extract(self._h, _p(f), _p(pcm), _p(hp))
        return hp

    def extract_batch(self, filters, pcm, n_threads=1):
        f = _c(filters, np.float32)
        pcm = _c(pcm, np.int16)
        n_clips = pcm.shape[0]
        assert pcm.shape == (n_clips, self.n_samples)
        hp = np.zeros((n_clips, self.n_hp), np.uint64)
        lib().hpfw_oracle_extract_batch(self._h, _p(f), _p(pcm), n_clips, _p(hp), int(n_threads))
        return hp


def db(mag):
    mag = _c(mag, np.float32)
    out = np.zeros_like(mag)
    lib().hpfw_oracle_db(_p(mag), mag.size, _p(out))
    return out


def project(filters, s_db):
    """filters: flat column-major 64 x 2420 (element (r,k) at r + 64 k); s_db [121][C]."""
    f = _c(filters, np.float32)
    s = _c(s_db, np.float32)
    c = s.shape[1]
    out = np.zeros((NFILT, c - CTX + 1), np.float32)
    lib().hpfw_oracle_project(_p(f), _p(s), c, _p(out))
    return out


def project_q(filters, s_db):
    """S9q: the projection in fixed point (exact int64 sums of once-rounded 24-bit factors)"""
    f = _c(filters, np.float32)
    s = _c(s_db, np.float32)
    c = s.shape[1]
    out = np.zeros((NFILT, c - CTX + 1), np.int64)
    lib().hpfw_oracle_project_q(_p(f), _p(s), c, _p(out))
    return out


def pack_q(proj):
    pr = _c(proj, np.int64)
    nf = pr.shape[1]
    hp = np.zeros(max(nf - LAG, 0), np.uint64)
    lib().hpfw_oracle_pack_q(_p(pr), nf, _p(hp))
    return hp


def delta_q(filters, s_db):
    """S9q with the lag-80 difference taken first: D[r][i] = Pq[r][i] - Pq[r][i + 80], int64 [64][c - 99]"""
    f = _c(filters, np.float32)
    s = _c(s_db, np.float32)
    c = s.shape[1]
    out = np.zeros((NFILT, max(c - CTX + 1 - LAG, 0)), np.int64)
    lib().hpfw_oracle_delta_q(_p(f), _p(s), c, _p(out))
    return out


def quantise_db(s_db):
    s = _c(s_db, np.float32)
    out = np.zeros(s.shape, np.int32)
    lib().hpfw_oracle_quantise_db(_p(s), s.size, _p(out))
    return out


def quantise_filters(filters):
    f = _c(filters, np.float32)
    out = np.zeros((NFILT, BINS * CTX), np.int32)
    lib().hpfw_oracle_quantise_filters(_p(f), _p(out))
    return out


def hashprints_from_db(filters, s_db):
    """dB spectrogram [121][C] -> hashprints with the projection in force (set_projection)"""
    return pack_q(project_q(filters, s_db)) if get_projection() else pack(project(filters, s_db))


def set_projection(mode):
    """what Plan.extract* use: 1 (default) = fixed point (S9q), 0 = the f32 fma chain (S9)"""
    lib().hpfw_oracle_set_projection(int(mode))


def get_projection():
    return int(lib().hpfw_oracle_get_projection())


def pack(proj):
    pr = _c(proj, np.float32)
    nf = pr.shape[1]
    hp = np.zeros(max(nf - LAG, 0), np.uint64)
    lib().hpfw_oracle_pack(_p(pr), nf, _p(hp))
    return hp


def hashprints_cfg(filters, s, context, lag, bits, return_projection=False):
    """HashprintHandle<uintN, SH, context, lag> on one spectrogram s [rows][cols]; filters flat column-major
    [bits][rows * context].  Returns uint64 words holding the `bits`-bit hashprints."""
    f = _c(filters, np.float32)
    s = _c(s, np.float32)
    rows, cols = s.shape
    nf = cols - context + 1
    if nf <= 0:
        return (np.zeros(0, np.uint64), np.zeros((bits, 0), np.float32)) if return_projection else np.zeros(0, np.uint64)
    proj = np.zeros((bits, nf), np.float32)
    lib().hpfw_oracle_project_cfg(_p(f), _p(s), rows, context, bits, cols, cols, _p(proj))
    hp = np.zeros(max(nf - lag, 0), np.uint64)
    lib().hpfw_oracle_pack_cfg(_p(proj), bits, lag, nf, nf, _p(hp))
    return (hp, proj) if return_projection else hp


def match_clip(q, r):
    q = _c(q, np.uint64)
    r = _c(r, np.uint64)
    d = ctypes.c_uint64(0)
    o = ctypes.c_int64(0)
    lib().hpfw_oracle_match_clip(_p(q), q.size, _p(r), r.size, ctypes.byref(d), ctypes.byref(o))
    return int(d.value), int(o.value)


def search_topk(db_hp, db_off, q_hp, q_off, topk, n_threads=1):
    db_hp = _c(db_hp, np.uint64)
    db_off = _c(db_off, np.int64)
    q_hp = _c(q_hp, np.uint64)
    q_off = _c(q_off, np.int64)
    n_q = q_off.size - 1
    out = np.zeros((n_q, topk), HIT_DTYPE)
    lib().hpfw_oracle_search_topk(_p(db_hp), _p(db_off), db_off.size - 1, _p(q_hp), _p(q_off), n_q,
                                  int(topk), _p(out), int(n_threads))
    return out


class Mel:
    """f3: MelSpectrogram<44100, 33, 4410, 441> (mel.h:34-104)"""

    def __init__(self):
        self._h = lib().hpfw_oracle_mel_create()

    def __del__(self):
        try:
            lib().hpfw_oracle_mel_destroy(self._h)
        except Exception:
            pass

    def tables(self):
        w = np.zeros(4410, np.float32)
        c = np.zeros((33, 2206), np.float32)
        lib().hpfw_oracle_mel_tables(self._h, _p(w), _p(c))
        return w, c

    @staticmethod
    def frames(n):
        return int(lib().hpfw_oracle_mel_frames(int(n)))

    def power(self, pcm):
        pcm = _c(pcm, np.int16)
        nfr = self.frames(pcm.size)
        p = np.zeros((33, nfr), np.float32)
        k = np.zeros(nfr, np.uint8)
        lib().hpfw_oracle_mel_power(self._h, _p(pcm), pcm.size, _p(p), _p(k))
        return p, k.astype(bool)

    def spectrogram(self, pcm):
        """dB-mel spectrogram [33][kept columns]"""
        pcm = _c(pcm, np.int16)
        nfr = self.frames(pcm.size)
        out = np.zeros((33, nfr), np.float32)
        c = lib().hpfw_oracle_mel_spectrogram(self._h, _p(pcm), pcm.size, _p(out))
        return np.ascontiguousarray(out[:, :c])


VOTE_DTYPE = np.dtype([("clip", "<i8"), ("offset", "<i8"), ("cnt", "<f4"), ("pad", "<f4")])


def knn_windows(db_hp, db_off, q_hp, win=64, nn=5):
    """exact nearest windows of every query position: keys [k - win + 1][nn] = dist << 40 | global position"""
    db_hp = _c(db_hp, np.uint64)
    db_off = _c(db_off, np.int64)
    q_hp = _c(q_hp, np.uint64)
    n_win = max(q_hp.size - win + 1, 0)
    keys = np.full((n_win, nn), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
    if n_win:
        lib().hpfw_oracle_knn_windows(_p(db_hp), _p(db_off), db_off.size - 1, _p(q_hp), q_hp.size, int(win), int(nn),
                                      _p(keys))
    return keys


def vote_windows(keys, db_off):
    """annoy_storage.h:45-61 over exact neighbours: the winning (clip, offset, count)"""
    keys = _c(keys, np.uint64)
    db_off = _c(db_off, np.int64)
    out = np.zeros(1, VOTE_DTYPE)
    nn = keys.shape[1] if keys.ndim == 2 else 5
    lib().hpfw_oracle_vote_windows(_p(keys), keys.shape[0], int(nn), _p(db_off), db_off.size - 1, _p(out))
    return out[0]


def search_votes(db_hp, db_off, q_hp, win=64, nn=5):
    return vote_windows(knn_windows(db_hp, db_off, q_hp, win, nn), db_off)


def log10(x):
    return lib().hpfw_oracle_log10(float(x))


def twiddle(m, n):
    re = ctypes.c_float(0)
    im = ctypes.c_float(0)
    lib().hpfw_oracle_twiddle(int(m), int(n), ctypes.byref(re), ctypes.byref(im))
    return re.value, im.value


def fft_dif(a, radix):
    a = np.array(a, dtype=np.complex64).copy()
    r = np.asarray(radix, np.int32)
    lib().hpfw_oracle_fft_dif(_p(a), a.size, _p(r), r.size)
    return a


def fft_idit(a, radix):
    a = np.array(a, dtype=np.complex64).copy()
    r = np.asarray(radix, np.int32)
    lib().hpfw_oracle_fft_idit(_p(a), a.size, _p(r), r.size)
    return a


def digit_pos(k, n, radix):
    r = np.asarray(radix, np.int32)
    return int(lib().hpfw_oracle_digit_pos(int(k), int(n), _p(r), r.size))
