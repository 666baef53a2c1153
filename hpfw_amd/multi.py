"""ctypes binding of libhpfw_gpu_multi.so (include/hpfw_gpu_multi.h): the native multi-GPU host path -- one
process, one handle per shard, one RCCL all-gather of the per-shard top-k lists per search.  (bench.py's N > 1
leg runs one process per GPU under torch.distributed instead, as its launch contract demands; both use the
same shard arithmetic and the same merge.)"""
import ctypes
import os

import numpy as np

from . import _lib

LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhpfw_gpu_multi.so")

EXPORTS = (
    "hpfw_gpu_group_create", "hpfw_gpu_group_create_env", "hpfw_gpu_group_destroy", "hpfw_gpu_group_size",
    "hpfw_gpu_group_handle", "hpfw_gpu_group_exchange", "hpfw_gpu_group_set_filters",
    "hpfw_gpu_group_extract_pcm16", "hpfw_gpu_group_index_build", "hpfw_gpu_group_index_size",
    "hpfw_gpu_shard_range", "hpfw_gpu_group_search_topk", "hpfw_gpu_group_cov_reset",
    "hpfw_gpu_group_cov_accumulate_pcm16", "hpfw_gpu_group_learn_filters",
    "hpfw_gpu_group_load", "hpfw_gpu_group_save", "hpfw_gpu_group_prepare", "hpfw_gpu_group_calc_hashprint",
)

_multi = None


def lib():
    global _multi
    if _multi is not None:
        return _multi
    _lib.lib()                                             # libhpfw_gpu.so first (the multi library is built on it)
    if not os.path.exists(LIB_PATH):
        raise _lib.HpfwError(f"{LIB_PATH} is missing: build it with hpfw_amd.build.build()")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    L.hpfw_gpu_group_create.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.hpfw_gpu_group_create_env.argtypes = [ctypes.POINTER(vp)]
    L.hpfw_gpu_group_destroy.argtypes = [vp]
    L.hpfw_gpu_group_destroy.restype = None
    L.hpfw_gpu_group_size.argtypes = [vp]
    L.hpfw_gpu_group_handle.argtypes = [vp, i32]
    L.hpfw_gpu_group_handle.restype = vp
    L.hpfw_gpu_group_exchange.argtypes = [vp]
    L.hpfw_gpu_group_exchange.restype = ctypes.c_char_p
    L.hpfw_gpu_group_set_filters.argtypes = [vp, vp]
    L.hpfw_gpu_group_extract_pcm16.argtypes = [vp, vp, i64, i64, vp]
    L.hpfw_gpu_group_index_build.argtypes = [vp, vp, vp, i64]
    L.hpfw_gpu_group_index_size.argtypes = [vp]
    L.hpfw_gpu_group_index_size.restype = i64
    L.hpfw_gpu_shard_range.argtypes = [i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.hpfw_gpu_shard_range.restype = None
    L.hpfw_gpu_group_search_topk.argtypes = [vp, vp, vp, i64, i32, vp]
    L.hpfw_gpu_group_cov_reset.argtypes = [vp]
    L.hpfw_gpu_group_cov_accumulate_pcm16.argtypes = [vp, vp, i64, i64]
    L.hpfw_gpu_group_learn_filters.argtypes = [vp, vp]
    L.hpfw_gpu_group_load.argtypes = [vp, ctypes.c_char_p]
    L.hpfw_gpu_group_save.argtypes = [vp, ctypes.c_char_p]
    L.hpfw_gpu_group_prepare.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32, ctypes.POINTER(i32)]
    L.hpfw_gpu_group_prepare.restype = ctypes.POINTER(_lib.FilenameHashprintPair)
    L.hpfw_gpu_group_calc_hashprint.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32)]
    L.hpfw_gpu_group_calc_hashprint.restype = ctypes.POINTER(ctypes.c_uint64)
    _multi = L
    return L


def shard_range(n_clips, shard, n_shards):
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    lib().hpfw_gpu_shard_range(int(n_clips), int(shard), int(n_shards), ctypes.byref(lo), ctypes.byref(hi))
    return int(lo.value), int(hi.value)


class GpuGroup:
    """devices: one ordinal per shard (an ordinal may repeat); None = HPFW_GPU_DEVICES / every visible device"""

    def __init__(self, devices=None):
        self._g = ctypes.c_void_p()
        if devices is None:
            _lib.check(lib().hpfw_gpu_group_create_env(ctypes.byref(self._g)))
        else:
            d = np.ascontiguousarray(devices, np.int32)
            _lib.check(lib().hpfw_gpu_group_create(_lib._hp(d), d.size, ctypes.byref(self._g)))

    def close(self):
        if getattr(self, "_g", None):
            lib().hpfw_gpu_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shards(self):
        return int(lib().hpfw_gpu_group_size(self._g))

    def handle(self, shard):
        """the hpfw_gpu handle of a shard (owned by the group): wrap it with hpfw_amd.Gpu.from_handle"""
        h = lib().hpfw_gpu_group_handle(self._g, int(shard))
        if not h:
            raise _lib.HpfwError(f"no shard {shard}")
        return h

    @property
    def exchange(self):
        return lib().hpfw_gpu_group_exchange(self._g).decode()

    def set_filters(self, filters_colmajor):
        f = np.ascontiguousarray(filters_colmajor, np.float32).ravel()
        assert f.size == 64 * 2420
        _lib.check(lib().hpfw_gpu_group_set_filters(self._g, _lib._hp(f)))

    def extract(self, pcm, n_hp):
        pcm = np.ascontiguousarray(pcm, np.int16)
        hp = np.zeros((pcm.shape[0], n_hp), np.uint64)
        _lib.check(lib().hpfw_gpu_group_extract_pcm16(self._g, _lib._hp(pcm), pcm.shape[1], pcm.shape[0], _lib._hp(hp)))
        return hp

    def index_build(self, hp, offsets):
        hp = np.ascontiguousarray(hp, np.uint64).ravel()
        off = np.ascontiguousarray(offsets, np.int64)
        _lib.check(lib().hpfw_gpu_group_index_build(self._g, _lib._hp(hp), _lib._hp(off), off.size - 1))

    def search_topk(self, q_hp, q_off, k):
        q = np.ascontiguousarray(q_hp, np.uint64).ravel()
        off = np.ascontiguousarray(q_off, np.int64)
        out = np.zeros((off.size - 1, k), _lib.HIT_DTYPE)
        _lib.check(lib().hpfw_gpu_group_search_topk(self._g, _lib._hp(q), _lib._hp(off), off.size - 1, int(k), _lib._hp(out)))
        return out

    def cov_reset(self):
        _lib.check(lib().hpfw_gpu_group_cov_reset(self._g))

    def cov_accumulate(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        _lib.check(lib().hpfw_gpu_group_cov_accumulate_pcm16(self._g, _lib._hp(pcm), pcm.shape[1], pcm.shape[0]))

    def learn_filters(self):
        f = np.zeros(64 * 2420, np.float32)
        _lib.check(lib().hpfw_gpu_group_learn_filters(self._g, _lib._hp(f)))
        return f

    # ---- ParallelCollector over the shards ------------------------------------------------------------
    def load(self, cache=""):
        _lib.check(lib().hpfw_gpu_group_load(self._g, cache.encode("utf-8")))

    def save(self, cache=""):
        _lib.check(lib().hpfw_gpu_group_save(self._g, cache.encode("utf-8")))

    def prepare(self, filenames):
        """list of (uint64 array, stem): this call's files in input order, then the older tracks of the cache"""
        raw = [str(f).encode("utf-8") for f in filenames]
        names = (ctypes.c_char_p * len(raw))(*raw)
        got = ctypes.c_int(0)
        res = lib().hpfw_gpu_group_prepare(self._g, names, len(raw), ctypes.byref(got))
        if not res:
            raise _lib.HpfwError("group prepare failed: " + _lib.lib().hpfw_gpu_last_error().decode())
        out = []
        for i in range(got.value):
            a = np.ctypeslib.as_array(res[i].hashprint, shape=(res[i].hp_size,)).astype(np.uint64).copy()
            out.append((a, res[i].filename.decode("utf-8")))
        _lib.lib().prepare_result_free(res, got)
        return out

    def calc_hashprint(self, filename):
        size = ctypes.c_int(0)
        hp = lib().hpfw_gpu_group_calc_hashprint(self._g, str(filename).encode("utf-8"), ctypes.byref(size))
        if not hp:
            raise _lib.HpfwError("group calc_hashprint failed: " + _lib.lib().hpfw_gpu_last_error().decode())
        a = np.ctypeslib.as_array(hp, shape=(size.value,)).astype(np.uint64).copy()
        _lib.lib().calc_hashprint_result_free(hp)
        return a
