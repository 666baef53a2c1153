"""Python twin of the reference's ctypes class (modules/python/pyhpfw/pyhpfw.py:13-81), bound to
the same eight C symbols (modules/python/parallel_collector_wrapper.hpp:21-38) that
libhpfw_gpu.so re-exports on top of the GPU path."""
import ctypes
from typing import List, Tuple

import numpy as np

from . import _lib


class ParallelCollector:
    def __init__(self):
        self.__lib = _lib.lib()
        self.__collector = self.__lib.par_collector_new()
        if not self.__collector:
            raise _lib.HpfwError("par_collector_new failed: " + self.__lib.hpfw_gpu_last_error().decode())

    @staticmethod
    def _c_strings(filenames):
        """a C array of NUL-terminated UTF-8 paths"""
        raw = [str(f).encode("utf-8") for f in filenames]
        return (ctypes.c_char_p * len(raw))(*raw)

    def prepare(self, filenames) -> List[Tuple[np.ndarray, str]]:
        """pyhpfw.py:46-61: list of (uint64 array, stem) -- array first, as the reference returns."""
        names = self._c_strings(filenames)
        n_got = ctypes.c_int(0)
        hps = self.__lib.par_collector_prepare(self.__collector, names, len(names), ctypes.byref(n_got))
        if not hps or (n_got.value == 0 and len(names)):
            # eigen-solve failure, out of memory, or every file unreadable: the C++ facade throws here too
            why = self.__lib.hpfw_gpu_last_error().decode()
            if hps:
                self.__lib.prepare_result_free(hps, 0)
            raise _lib.HpfwError("prepare failed: " + why)
        got = n_got
        out = []
        for h in range(got.value):
            n = hps[h].hp_size
            a = np.ctypeslib.as_array(hps[h].hashprint, shape=(n,)).astype(np.uint64).copy()
            out.append((a, hps[h].filename.decode("utf-8")))
        self.__lib.prepare_result_free(hps, got)
        return out

    def calc_hashprint(self, filename: str) -> np.ndarray:
        """pyhpfw.py:63-72"""
        size = ctypes.c_int(0)
        hp = self.__lib.par_collector_calc_hashprint(self.__collector, filename.encode("utf-8"),
                                                     ctypes.byref(size))
        if not hp:
            raise _lib.HpfwError(f"calc_hashprint({filename!r}) failed: "
                                 + self.__lib.hpfw_gpu_last_error().decode())
        a = np.ctypeslib.as_array(hp, shape=(size.value,)).astype(np.uint64).copy()
        self.__lib.calc_hashprint_result_free(hp)
        return a

    def calc_hashprints(self, filenames) -> List[Tuple[np.ndarray, str]]:
        """calc_hashprint for many files in one batched call (not in the reference's class): a list of
        (uint64 array or None for a file that failed, stem), one entry per file, in input order"""
        pyarr = self._c_strings(filenames)
        hps = self.__lib.par_collector_calc_hashprints(self.__collector, pyarr, len(pyarr))
        if not hps:
            raise _lib.HpfwError("calc_hashprints failed: " + self.__lib.hpfw_gpu_last_error().decode())
        out = []
        for h in range(len(pyarr)):
            a = None
            if hps[h].hashprint:
                a = np.ctypeslib.as_array(hps[h].hashprint, shape=(hps[h].hp_size,)).astype(np.uint64).copy()
            out.append((a, hps[h].filename.decode("utf-8")))
        self.__lib.prepare_result_free(hps, len(pyarr))
        return out

    def load(self, cache: str = ""):
        self.__lib.par_collector_load(self.__collector, cache.encode("utf-8"))

    def save(self, cache: str = ""):
        self.__lib.par_collector_save(self.__collector, cache.encode("utf-8"))

    def __del__(self):
        if getattr(self, "_ParallelCollector__collector", None):
            self.__lib.par_collector_del(self.__collector)
            self.__collector = None
