"""Times the host eigen-solve (2420 x 2420, 64 leading eigenvectors) for several team sizes."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpfw_amd  # noqa: E402

L = hpfw_amd.lib()
rng = np.random.default_rng(1)
n, m = 2420, 64
x = rng.standard_normal((n, 3000)) * np.linspace(5, 0.1, n)[:, None]
a = (x @ x.T / 3000).astype(np.float32)
for th in sys.argv[1:] or ["1", "8", "32"]:
    os.environ["HPFW_EIGEN_THREADS"] = th
    out = np.zeros((m, n), np.float32)
    t0 = time.perf_counter()
    L.hpfw_gpu_host_top_eigenvectors(a.ctypes.data_as(ctypes.c_void_p), n, m, out.ctypes.data_as(ctypes.c_void_p), None)
    print(th, "threads:", round(time.perf_counter() - t0, 3), "s")
