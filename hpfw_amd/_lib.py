"""ctypes binding of libhpfw_gpu.so (include/hpfw_gpu.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the symbols
raises.  The library is built in-tree by hpfw_amd.build.build() (hipcc --offload-arch=gfx950).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HPFW_GPU_LIB: another build of the same library, for diagnosis -- tools/interfere.py)
LIB_PATH = os.environ.get("HPFW_GPU_LIB") or os.path.join(_HERE, "lib", "libhpfw_gpu.so")

HIT_DTYPE = np.dtype([("dist", "<u4"), ("clip", "<u4"), ("offset", "<i4"), ("pad", "<u4")])
VOTE_DTYPE = np.dtype([("clip", "<u4"), ("pad", "<u4"), ("offset", "<i8"), ("cnt", "<f4"), ("pad2", "<f4")])

KERNEL_KINDS = ("fwd_rows", "fwd_cols", "cq_chirpz", "db", "project_mfma", "delta_pack",
                "hamming_scan", "topk", "pcm_pairs", "fwd_span")

# every symbol include/hpfw_gpu.h declares (tests check that the library exports all of them)
EXPORTS = (
    "hpfw_gpu_last_error", "hpfw_gpu_version", "hpfw_gpu_create", "hpfw_gpu_destroy", "hpfw_gpu_device",
    "hpfw_gpu_set_filters", "hpfw_gpu_geometry", "hpfw_gpu_extract_pcm16",
    "hpfw_gpu_extract_pcm16_host", "hpfw_gpu_set_batch", "hpfw_gpu_stage_spectrum",
    "hpfw_gpu_stage_cqmag", "hpfw_gpu_stage_db", "hpfw_gpu_stage_project", "hpfw_gpu_stage_pack",
    "hpfw_gpu_cov_reset", "hpfw_gpu_cov_accumulate_pcm16", "hpfw_gpu_cov_accumulate_pcm16_host",
    "hpfw_gpu_cov_accumulate_db", "hpfw_gpu_cov_get",
    "hpfw_gpu_cov_set", "hpfw_gpu_learn_filters", "hpfw_gpu_host_top_eigenvectors",
    "hpfw_gpu_cov_device", "hpfw_gpu_cov_files", "hpfw_gpu_cov_set_files",
    "hpfw_gpu_index_clear", "hpfw_gpu_index_add", "hpfw_gpu_index_add_device",
    "hpfw_gpu_index_size", "hpfw_gpu_index_set_clip_base", "hpfw_gpu_search_topk_device",
    "hpfw_gpu_search_topk", "hpfw_gpu_merge_topk", "hpfw_gpu_timer_start", "hpfw_gpu_timer_stop",
    "hpfw_gpu_index_get", "hpfw_gpu_extract_db_host", "hpfw_gpu_stage_spectrogram",
    "hpfw_gpu_search_votes", "hpfw_gpu_knn_windows", "hpfw_gpu_supported_length",
    "hpfw_gpu_mel_frames", "hpfw_gpu_mel_spectrogram_pcm16", "hpfw_gpu_mel_spectrogram_pcm16_host",
    "hpfw_gpu_cfg_set_filters", "hpfw_gpu_cfg_hashprints", "hpfw_gpu_mel_hashprints_pcm16_host",
    "hpfw_gpu_cfg_cov_reset", "hpfw_gpu_cfg_cov_accumulate", "hpfw_gpu_cfg_cov_get", "hpfw_gpu_cfg_learn_filters",
    "hpfw_gpu_set_kernel_timing", "hpfw_gpu_get_kernel_timing", "hpfw_gpu_plan_checksum",
    "hpfw_gpu_plan_checksum_ex", "hpfw_gpu_set_conventions", "hpfw_gpu_chirpz_table", "hpfw_gpu_debug_workspace", "hpfw_gpu_prepare_length", "hpfw_gpu_set_projection", "hpfw_gpu_get_projection",
    "hpfw_gpu_hashprints_from_db", "hpfw_gpu_stage_delta_q",
    "par_collector_new", "par_collector_del", "par_collector_prepare",
    "par_collector_calc_hashprint", "par_collector_calc_hashprints", "par_collector_save", "par_collector_load",
    "prepare_result_free", "calc_hashprint_result_free",
)


class HandleConfig(ctypes.Structure):
    """hpfw_handle_config: HashprintHandle<N, SH, FramesContext, T> (hashprint_handle.h:50-64)"""
    _fields_ = [("rows", ctypes.c_int32), ("context", ctypes.c_int32), ("lag", ctypes.c_int32), ("bits", ctypes.c_int32)]


COMBINER_CONFIG = (33, 32, 50, 16)      # combiner.h:12: HashPrint<uint16_t, MelSpectrogram<>, 32, 50>


class Geometry(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("n_samples", "n1", "n2", "kmin", "kmax", "m", "c", "n_frames", "n_hp")]


class FilenameHashprintPair(ctypes.Structure):
    """modules/python/pyhpfw/pyhpfw.py:7-10 of the reference."""
    _fields_ = [("filename", ctypes.c_char_p),
                ("hashprint", ctypes.POINTER(ctypes.c_uint64)),
                ("hp_size", ctypes.c_int)]


class HpfwError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the HIP library, or raise: the product has no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HpfwError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32
    L.hpfw_gpu_last_error.restype = ctypes.c_char_p
    L.hpfw_gpu_version.restype = ctypes.c_char_p
    L.hpfw_gpu_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.hpfw_gpu_destroy.argtypes = [vp]
    L.hpfw_gpu_destroy.restype = None
    L.hpfw_gpu_set_filters.argtypes = [vp, vp]
    L.hpfw_gpu_geometry.argtypes = [vp, i64, ctypes.POINTER(Geometry)]
    L.hpfw_gpu_extract_pcm16.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_extract_pcm16_host.argtypes = [vp, vp, i64, i64, vp]
    L.hpfw_gpu_set_batch.argtypes = [vp, i32]
    L.hpfw_gpu_stage_spectrum.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_stage_cqmag.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_stage_db.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_stage_project.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_stage_pack.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_cov_reset.argtypes = [vp]
    L.hpfw_gpu_cov_accumulate_pcm16.argtypes = [vp, vp, i64, i64, vp]
    L.hpfw_gpu_cov_accumulate_pcm16_host.argtypes = [vp, vp, i64, i64]
    L.hpfw_gpu_cov_accumulate_db.argtypes = [vp, vp, i64, i64, vp]
    L.hpfw_gpu_cov_get.argtypes = [vp, vp, ctypes.POINTER(i64)]
    L.hpfw_gpu_cov_set.argtypes = [vp, vp, i64]
    L.hpfw_gpu_learn_filters.argtypes = [vp, vp]
    L.hpfw_gpu_host_top_eigenvectors.argtypes = [vp, i32, i32, vp, vp]
    L.hpfw_gpu_index_clear.argtypes = [vp]
    L.hpfw_gpu_index_add.argtypes = [vp, vp, vp, i64]
    L.hpfw_gpu_index_add_device.argtypes = [vp, vp, vp, i64, vp]
    L.hpfw_gpu_index_size.argtypes = [vp]
    L.hpfw_gpu_index_size.restype = i64
    L.hpfw_gpu_index_set_clip_base.argtypes = [vp, u32]
    L.hpfw_gpu_search_topk_device.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.hpfw_gpu_search_topk.argtypes = [vp, vp, vp, i64, i32, vp]
    L.hpfw_gpu_merge_topk.argtypes = [vp, i32, i64, i32, vp]
    L.hpfw_gpu_timer_start.argtypes = [vp, vp]
    L.hpfw_gpu_timer_stop.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.hpfw_gpu_stage_spectrogram.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_mel_frames.argtypes = [i64]
    L.hpfw_gpu_mel_frames.restype = i64
    L.hpfw_gpu_mel_spectrogram_pcm16.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    L.hpfw_gpu_mel_spectrogram_pcm16_host.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_cfg_set_filters.argtypes = [vp, ctypes.POINTER(HandleConfig), vp]
    L.hpfw_gpu_cfg_hashprints.argtypes = [vp, ctypes.POINTER(HandleConfig), vp, vp, i64, i64, vp, i64, vp, vp]
    L.hpfw_gpu_mel_hashprints_pcm16_host.argtypes = [vp, vp, i64, i64, vp, i64, vp]
    L.hpfw_gpu_cfg_cov_reset.argtypes = [vp, ctypes.POINTER(HandleConfig)]
    L.hpfw_gpu_cfg_cov_accumulate.argtypes = [vp, ctypes.POINTER(HandleConfig), vp, vp, i64, i64, vp]
    L.hpfw_gpu_cfg_cov_get.argtypes = [vp, ctypes.POINTER(HandleConfig), vp, ctypes.POINTER(i64)]
    L.hpfw_gpu_cfg_learn_filters.argtypes = [vp, ctypes.POINTER(HandleConfig), vp]
    L.hpfw_gpu_supported_length.argtypes = [i64]
    L.hpfw_gpu_supported_length.restype = i64
    L.hpfw_gpu_search_votes.argtypes = [vp, vp, vp, i64, vp]
    L.hpfw_gpu_knn_windows.argtypes = [vp, vp, vp, i64, vp, i64]
    L.hpfw_gpu_index_get.argtypes = [vp, vp, vp, ctypes.c_int64]
    L.hpfw_gpu_extract_db_host.argtypes = [vp, vp, i32, i32, vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.hpfw_gpu_set_kernel_timing.argtypes = [vp, i32]
    L.hpfw_gpu_get_kernel_timing.argtypes = [vp, vp, vp, vp, ctypes.POINTER(i32)]
    L.hpfw_gpu_plan_checksum.argtypes = [i64, vp]
    L.hpfw_gpu_plan_checksum_ex.argtypes = [i64, i32, u32, vp]
    L.hpfw_gpu_chirpz_table.argtypes = [vp, i64, i32, vp, i64, vp]
    L.hpfw_gpu_debug_workspace.argtypes = [vp, i32, vp, vp]
    L.hpfw_gpu_prepare_length.argtypes = [vp, i64]
    L.hpfw_gpu_set_projection.argtypes = [vp, i32]
    L.hpfw_gpu_get_projection.argtypes = [vp]
    L.hpfw_gpu_hashprints_from_db.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hpfw_gpu_stage_delta_q.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    L.hpfw_gpu_set_conventions.argtypes = [vp, u32]
    L.par_collector_new.restype = vp
    L.par_collector_del.argtypes = [vp]
    L.par_collector_del.restype = None
    L.par_collector_prepare.restype = ctypes.POINTER(FilenameHashprintPair)
    L.par_collector_prepare.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32, ctypes.POINTER(i32)]
    L.par_collector_calc_hashprints.restype = ctypes.POINTER(FilenameHashprintPair)
    L.par_collector_calc_hashprints.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32]
    L.par_collector_calc_hashprint.restype = ctypes.POINTER(ctypes.c_uint64)
    L.par_collector_calc_hashprint.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32)]
    L.par_collector_load.argtypes = [vp, ctypes.c_char_p]
    L.par_collector_load.restype = None
    L.par_collector_save.argtypes = [vp, ctypes.c_char_p]
    L.par_collector_save.restype = None
    L.prepare_result_free.argtypes = [vp, i32]
    L.prepare_result_free.restype = None
    L.calc_hashprint_result_free.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.calc_hashprint_result_free.restype = None
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise HpfwError(f"hpfw_gpu error {rc}: {lib().hpfw_gpu_last_error().decode()}")


def _hp(a):
    """host pointer of a contiguous numpy array"""
    return a.ctypes.data_as(ctypes.c_void_p)


class Gpu:
    """One handle = one MI355X device.  Device-pointer methods take integers (tensor.data_ptr())."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        check(lib().hpfw_gpu_create(int(device), ctypes.byref(self._h)))

    @classmethod
    def from_handle(cls, handle):
        """a view of a handle somebody else owns (a shard of hpfw_amd.multi.GpuGroup): close() does not destroy it"""
        g = cls.__new__(cls)
        g._h = ctypes.c_void_p(handle) if not isinstance(handle, ctypes.c_void_p) else handle
        g._borrowed = True
        return g

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                lib().hpfw_gpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:        # interpreter teardown: the module globals may already be gone
            pass

    # ---- configuration -------------------------------------------------------------------
    def set_filters(self, filters_colmajor):
        f = np.ascontiguousarray(filters_colmajor, np.float32).ravel()
        if f.size != 64 * 2420:
            raise ValueError("filters must hold 64 x 2420 floats (column-major)")
        check(lib().hpfw_gpu_set_filters(self._h, _hp(f)))

    def geometry(self, n_samples):
        g = Geometry()
        check(lib().hpfw_gpu_geometry(self._h, int(n_samples), ctypes.byref(g)))
        return g

    def set_conventions(self, flags):
        """HPFW_CONV_* bits: essentia conventions that cannot be checked offline (include/hpfw_gpu.h)"""
        check(lib().hpfw_gpu_set_conventions(self._h, int(flags)))

    def set_batch(self, clips_per_pass):
        check(lib().hpfw_gpu_set_batch(self._h, int(clips_per_pass)))

    # ---- extraction ----------------------------------------------------------------------
    def extract_dev(self, d_pcm, n_samples, n_clips, d_hp, stream=0):
        check(lib().hpfw_gpu_extract_pcm16(self._h, d_pcm, n_samples, n_clips, d_hp, stream))

    def extract(self, pcm):
        """pcm: int16 [n_clips][n_samples] (host) -> uint64 [n_clips][n_hp]"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        g = self.geometry(pcm.shape[1])
        hp = np.zeros((pcm.shape[0], g.n_hp), np.uint64)
        check(lib().hpfw_gpu_extract_pcm16_host(self._h, _hp(pcm), pcm.shape[1], pcm.shape[0], _hp(hp)))
        return hp

    def set_projection(self, mode):
        """1 (default): fixed-point projection, exact integer sums (S9q); 0: the f32 fma chain (S9)"""
        check(lib().hpfw_gpu_set_projection(self._h, int(mode)))

    def get_projection(self):
        return int(lib().hpfw_gpu_get_projection(self._h))

    def hashprints_from_db_dev(self, d_db, n_clips, c, d_hp, stream=0):
        check(lib().hpfw_gpu_hashprints_from_db(self._h, d_db, n_clips, c, d_hp, stream))

    def stage_delta_q_dev(self, d_db, n_clips, c, d_delta, d_hp=0, stream=0):
        """the exact integer sums of the fixed-point projection, int64 [n_clips][64][c - 99] (parity checkpoint)"""
        check(lib().hpfw_gpu_stage_delta_q(self._h, d_db, n_clips, c, d_delta, d_hp, stream))

    def prepare_length(self, n_samples):
        """build the host half of the tables of a clip length on the calling thread (thread-safe; see hpfw_gpu.h)"""
        check(lib().hpfw_gpu_prepare_length(self._h, int(n_samples)))

    def debug_workspace(self, which):
        """(device pointer, bytes) of extraction workspace `which` as the last call left it (diagnosis)"""
        p, b = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().hpfw_gpu_debug_workspace(self._h, which, ctypes.byref(p), ctypes.byref(b)))
        return p.value or 0, b.value

    def chirpz_table(self, n_samples, which):
        """device-generated table as complex64: 0 w, 1 T_L, 2 Bhat, 3 w[k] / L of the chirp-z forward transform; 4 (every
        length) the constant-Q stage's windows, bands concatenated"""
        count = ctypes.c_int64(0)
        check(lib().hpfw_gpu_chirpz_table(self._h, n_samples, which, None, 0, ctypes.byref(count)))
        out = np.zeros(count.value, np.float32)
        check(lib().hpfw_gpu_chirpz_table(self._h, n_samples, which, _hp(out), count.value, ctypes.byref(count)))
        return out.view(np.complex64)

    def stage_spectrum_dev(self, d_pcm, n_samples, n_clips, d_x, stream=0):
        check(lib().hpfw_gpu_stage_spectrum(self._h, d_pcm, n_samples, n_clips, d_x, stream))

    def stage_cqmag_dev(self, d_x, n_samples, n_clips, d_mag, stream=0):
        check(lib().hpfw_gpu_stage_cqmag(self._h, d_x, n_samples, n_clips, d_mag, stream))

    def stage_db_dev(self, d_mag, n_clips, c, d_db, stream=0):
        check(lib().hpfw_gpu_stage_db(self._h, d_mag, n_clips, c, d_db, stream))

    def stage_project_dev(self, d_db, n_clips, c, d_proj, stream=0):
        check(lib().hpfw_gpu_stage_project(self._h, d_db, n_clips, c, d_proj, stream))

    def stage_pack_dev(self, d_proj, n_clips, n_frames, d_hp, stream=0):
        check(lib().hpfw_gpu_stage_pack(self._h, d_proj, n_clips, n_frames, d_hp, stream))

    def stage_spectrogram_dev(self, d_pcm, n_samples, n_clips, d_db, stream=0):
        check(lib().hpfw_gpu_stage_spectrogram(self._h, d_pcm, n_samples, n_clips, d_db, stream))

    # ---- Mel front-end ---------------------------------------------------------------------
    def mel_spectrogram(self, pcm):
        """pcm int16 [n_clips][n] (host) -> list of dB-mel spectrograms [33][kept columns] (mel.h:34-104)"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        nf = int(lib().hpfw_gpu_mel_frames(pcm.shape[1]))
        out = np.zeros((pcm.shape[0], 33, nf), np.float32)
        cols = np.zeros(pcm.shape[0], np.int32)
        check(lib().hpfw_gpu_mel_spectrogram_pcm16_host(self._h, _hp(pcm), pcm.shape[1], pcm.shape[0], _hp(out), _hp(cols)))
        return [np.ascontiguousarray(out[i, :, :cols[i]]) for i in range(pcm.shape[0])]

    # ---- HashprintHandle with other template arguments ---------------------------------------
    def cfg_set_filters(self, cfg, filters_colmajor):
        c = HandleConfig(*cfg)
        f = np.ascontiguousarray(filters_colmajor, np.float32).ravel()
        if f.size != c.bits * c.rows * c.context:
            raise ValueError("filters must hold bits x rows * context floats (column-major)")
        check(lib().hpfw_gpu_cfg_set_filters(self._h, ctypes.byref(c), _hp(f)))

    def cfg_hashprints_dev(self, cfg, d_s, d_cols, n_clips, stride, d_hp, hp_stride, d_proj=0, stream=0):
        c = HandleConfig(*cfg)
        check(lib().hpfw_gpu_cfg_hashprints(self._h, ctypes.byref(c), d_s, d_cols, n_clips, stride, d_hp, hp_stride,
                                            d_proj, stream))

    def cfg_cov_reset(self, cfg):
        check(lib().hpfw_gpu_cfg_cov_reset(self._h, ctypes.byref(HandleConfig(*cfg))))

    def cfg_cov_accumulate_dev(self, cfg, d_s, d_cols, n_clips, stride, stream=0):
        check(lib().hpfw_gpu_cfg_cov_accumulate(self._h, ctypes.byref(HandleConfig(*cfg)), d_s, d_cols, n_clips, stride, stream))

    def cfg_cov_get(self, cfg):
        kt = cfg[0] * cfg[1]
        cov = np.zeros((kt, kt), np.float32)
        n = ctypes.c_int64(0)
        check(lib().hpfw_gpu_cfg_cov_get(self._h, ctypes.byref(HandleConfig(*cfg)), _hp(cov), ctypes.byref(n)))
        return cov, int(n.value)

    def cfg_learn_filters(self, cfg):
        f = np.zeros(cfg[3] * cfg[0] * cfg[1], np.float32)
        check(lib().hpfw_gpu_cfg_learn_filters(self._h, ctypes.byref(HandleConfig(*cfg)), _hp(f)))
        return f

    def mel_hashprints(self, pcm):
        """the combiner's Algo (combiner.h:12) on host PCM [n_clips][n]: list of uint16 hashprint arrays"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        stride = max(int(lib().hpfw_gpu_mel_frames(pcm.shape[1])) - 81, 1)
        hp = np.zeros((pcm.shape[0], stride), np.uint16)
        n = np.zeros(pcm.shape[0], np.int32)
        check(lib().hpfw_gpu_mel_hashprints_pcm16_host(self._h, _hp(pcm), pcm.shape[1], pcm.shape[0], _hp(hp), stride, _hp(n)))
        return [hp[i, :n[i]].copy() for i in range(pcm.shape[0])]

    # ---- filter learning ------------------------------------------------------------------
    def cov_reset(self):
        check(lib().hpfw_gpu_cov_reset(self._h))

    def cov_accumulate_dev(self, d_pcm, n_samples, n_clips, stream=0):
        check(lib().hpfw_gpu_cov_accumulate_pcm16(self._h, d_pcm, n_samples, n_clips, stream))

    def cov_accumulate(self, pcm):
        """pcm: int16 [n_clips][n_samples] on the host"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        check(lib().hpfw_gpu_cov_accumulate_pcm16_host(self._h, _hp(pcm), pcm.shape[1], pcm.shape[0]))

    def cov_accumulate_db_dev(self, d_db, n_clips, c, stream=0):
        check(lib().hpfw_gpu_cov_accumulate_db(self._h, d_db, n_clips, c, stream))

    def cov_get(self):
        """(accum_cov [2420][2420] float32, number of clips accumulated)"""
        cov = np.zeros((2420, 2420), np.float32)
        n = ctypes.c_int64(0)
        check(lib().hpfw_gpu_cov_get(self._h, _hp(cov), ctypes.byref(n)))
        return cov, int(n.value)

    def cov_set(self, cov, n_files):
        c = np.ascontiguousarray(cov, np.float32)
        assert c.shape == (2420, 2420)
        check(lib().hpfw_gpu_cov_set(self._h, _hp(c), int(n_files)))

    def learn_filters(self):
        """eigen-solve the accumulated covariance, install and return the filters (flat column-major)"""
        f = np.zeros(64 * 2420, np.float32)
        check(lib().hpfw_gpu_learn_filters(self._h, _hp(f)))
        return f

    # ---- index + search ------------------------------------------------------------------
    def index_clear(self):
        check(lib().hpfw_gpu_index_clear(self._h))

    def index_add(self, hp, offsets):
        hp = np.ascontiguousarray(hp, np.uint64).ravel()
        off = np.ascontiguousarray(offsets, np.int64)
        check(lib().hpfw_gpu_index_add(self._h, _hp(hp), _hp(off), off.size - 1))

    def index_add_dev(self, d_hp, offsets, stream=0):
        off = np.ascontiguousarray(offsets, np.int64)
        check(lib().hpfw_gpu_index_add_device(self._h, d_hp, _hp(off), off.size - 1, stream))

    def index_get(self):
        """(hashprints uint64 [total], offsets int64 [n_clips + 1]) copied back from HBM"""
        off = np.zeros(self.index_size() + 1, np.int64)
        check(lib().hpfw_gpu_index_get(self._h, _hp(off), None, 0))
        hp = np.zeros(int(off[-1]), np.uint64)
        check(lib().hpfw_gpu_index_get(self._h, _hp(off), _hp(hp), hp.size))
        return hp, off

    def extract_db(self, s_colmajor):
        """hashprints of a cached dB spectrogram: s_colmajor float32 [cols][121] (Eigen column-major
        [121 x cols] as cache/spectros/<stem> holds it)"""
        s = np.ascontiguousarray(s_colmajor, np.float32)
        cols, rows = s.shape
        n = ctypes.c_int64(0)
        hp = np.zeros(max(cols - 99, 0), np.uint64)
        check(lib().hpfw_gpu_extract_db_host(self._h, _hp(s), rows, cols, _hp(hp), hp.size, ctypes.byref(n)))
        return hp[:n.value]

    def index_size(self):
        return int(lib().hpfw_gpu_index_size(self._h))

    def index_set_clip_base(self, base):
        check(lib().hpfw_gpu_index_set_clip_base(self._h, int(base)))

    def search_topk(self, q_hp, q_off, k):
        q = np.ascontiguousarray(q_hp, np.uint64).ravel()
        off = np.ascontiguousarray(q_off, np.int64)
        out = np.zeros((off.size - 1, k), HIT_DTYPE)
        check(lib().hpfw_gpu_search_topk(self._h, _hp(q), _hp(off), off.size - 1, int(k), _hp(out)))
        return out

    def search_votes(self, q_hp, q_off):
        """AnnStorage-style voting search with exact neighbours: VOTE_DTYPE [n_q]"""
        q = np.ascontiguousarray(q_hp, np.uint64).ravel()
        off = np.ascontiguousarray(q_off, np.int64)
        out = np.zeros(off.size - 1, VOTE_DTYPE)
        check(lib().hpfw_gpu_search_votes(self._h, _hp(q), _hp(off), off.size - 1, _hp(out)))
        return out

    def knn_windows(self, q_hp, q_off):
        """the 5 nearest 64-hashprint windows of every query position: keys [n_windows][5]"""
        q = np.ascontiguousarray(q_hp, np.uint64).ravel()
        off = np.ascontiguousarray(q_off, np.int64)
        n_win = int(np.maximum(np.diff(off) - 63, 0).sum())
        keys = np.zeros((n_win, 5), np.uint64)
        check(lib().hpfw_gpu_knn_windows(self._h, _hp(q), _hp(off), off.size - 1, _hp(keys), keys.size))
        return keys

    def search_topk_dev(self, d_q, q_off, k, d_out, stream=0):
        off = np.ascontiguousarray(q_off, np.int64)
        check(lib().hpfw_gpu_search_topk_device(self._h, d_q, _hp(off), off.size - 1, int(k), d_out, stream))

    # ---- timing --------------------------------------------------------------------------
    def timer_start(self, stream=0):
        check(lib().hpfw_gpu_timer_start(self._h, stream))

    def timer_stop(self, stream=0):
        ms = ctypes.c_float(0)
        check(lib().hpfw_gpu_timer_stop(self._h, stream, ctypes.byref(ms)))
        return float(ms.value)

    def set_kernel_timing(self, mask):
        check(lib().hpfw_gpu_set_kernel_timing(self._h, int(mask)))

    def kernel_timing(self):
        n = ctypes.c_int(16)
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        launches = (ctypes.c_int * 16)()
        check(lib().hpfw_gpu_get_kernel_timing(self._h, names, ms, launches, ctypes.byref(n)))
        return {names[i].decode(): (float(ms[i]), int(launches[i])) for i in range(n.value)}


def merge_topk(per_shard_hits, k):
    """per_shard_hits: [n_shards][n_q][k] HIT_DTYPE -> [n_q][k], ascending (dist, clip)."""
    a = np.ascontiguousarray(per_shard_hits, HIT_DTYPE)
    n_shards, n_q, kk = a.shape
    assert kk == k
    out = np.zeros((n_q, k), HIT_DTYPE)
    check(lib().hpfw_gpu_merge_topk(_hp(a), n_shards, n_q, k, _hp(out)))
    return out


def supported_length(n_samples):
    """the smallest supported clip length >= n_samples, or -1"""
    return int(lib().hpfw_gpu_supported_length(int(n_samples)))


CONV_HANN_PERIODIC, CONV_LG_HALF_EVEN, CONV_FLOAT_GEOMETRY, CONV_NO_IFFT_SCALE = 1, 2, 4, 8


def plan_checksum(n_samples, conventions=0):
    out = np.zeros(8, np.uint64)
    rc = lib().hpfw_gpu_plan_checksum_ex(abs(int(n_samples)), int(n_samples < 0), int(conventions), _hp(out))
    if rc != 0:
        raise HpfwError(f"unsupported clip length {n_samples}")
    return out
