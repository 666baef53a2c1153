"""hpfw_amd -- MI355X-native (gfx950) hashprint fingerprinting hot path of hpfw.

Only what the path needs: csrc/ (HIP kernels + the C-ABI of include/hpfw_gpu.h), a ctypes
binding, a twin of the reference's Python class (modules/python/pyhpfw/pyhpfw.py) and the
synthetic-audio generator used by tests and bench.py.  No CPU fallback exists.
"""
from ._lib import (Gpu, HpfwError, COMBINER_CONFIG, HIT_DTYPE, VOTE_DTYPE, KERNEL_KINDS, LIB_PATH, lib, merge_topk,  # noqa: F401
                   plan_checksum, supported_length)
from .collector import ParallelCollector  # noqa: F401
from .liveid import LiveSongIdentification  # noqa: F401

__all__ = ["Gpu", "HpfwError", "HIT_DTYPE", "VOTE_DTYPE", "KERNEL_KINDS", "LIB_PATH", "lib", "merge_topk",
           "plan_checksum", "supported_length", "ParallelCollector", "LiveSongIdentification"]
