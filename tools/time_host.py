"""PCIe-inclusive extraction rate: clips handed over as HOST int16 buffers through
hpfw_gpu_extract_pcm16_host, from pageable and from pinned memory (python3 tools/time_host.py [clips])."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpfw_amd  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(np.random.default_rng(0).standard_normal((2420, 64)).astype(np.float32))
n_hp = g.geometry(n).n_hp
L = hpfw_amd.lib()
pageable = torch.randint(-3000, 3000, (n_clips, n), dtype=torch.int16)
pinned = pageable.pin_memory()
hp = np.zeros((n_clips, n_hp), np.uint64)
for name, buf in (("pageable", pageable), ("pinned", pinned)):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        hpfw_amd._lib.check(L.hpfw_gpu_extract_pcm16_host(g._h, ctypes.c_void_p(buf.data_ptr()), n, n_clips,
                                                           hp.ctypes.data_as(ctypes.c_void_p)))
        best = min(best, time.perf_counter() - t0)
    print(f"host {name}: {n_clips} x 30 s clips in {best * 1e3:.1f} ms = {n_clips / best:.0f} clips/s "
          f"({n_clips * n * 2 / best / 1e9:.1f} GB/s of PCM)")
