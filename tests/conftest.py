import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Stage-by-stage parity first, the multi-process and stress tests last: with `-x` a failing stress test must not leave
# the tests that pin the reference's functions one by one unreached (round 3's record).
_ORDER = ("test_gpu_parity", "test_gpu_projection", "test_gpu_configs", "test_gpu_api", "test_gpu_learn", "test_gpu_fuzz",
          "test_gpu_dist", "test_gpu_multi", "test_gpu_shared")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1          # everything else (the CPU-side tests) first, as collected
    items.sort(key=rank)                                             # (stable: the order inside a file is kept)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only)."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def filters():
    from hpfw_amd import synth
    return synth.make_filters()


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this environment")
    return torch


@pytest.fixture(scope="session")
def gpu(torch_cuda, filters):
    """hpfw_amd.Gpu handle with the filter fixture loaded.  Fails loudly when the HIP library is
    missing -- there is no fallback."""
    import hpfw_amd
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    yield g
    g.close()


@pytest.fixture(params=["mfma", "shift", "popcount"])
def scan_path(request):
    """the three scan kernels: the fp4 matrix-core contraction over groups of 32 queries (default), its
    one-query variant with shifted rows (default below 8 queries) and the xor/popcount kernel"""
    import os
    var = {"mfma": "HPFW_SEARCH_MFMA", "shift": "HPFW_SEARCH_SHIFT", "popcount": "HPFW_SEARCH_POPC"}[request.param]
    os.environ[var] = "1"
    yield request.param
    os.environ.pop(var, None)


def bits_equal(a, b):
    """bitwise equality of two float32 arrays, +0 == -0 excluded on purpose"""
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    return a.shape == b.shape and bool((a == b).all())


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
