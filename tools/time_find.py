"""Latency of the host-query search entry point hpfw_gpu_search_topk for one query (GpuStorage::find) on a
small and on a larger index, and of the voting search (python3 tools/time_find.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import hpfw_amd  # noqa: E402

rng = np.random.default_rng(0)
g = hpfw_amd.Gpu(0)
for n_clips in (100, 10000):
    db = rng.integers(0, 2 ** 64, size=n_clips * 2320, dtype=np.uint64)
    g.index_clear()
    g.index_add(db, np.arange(0, (n_clips + 1) * 2320, 2320, dtype=np.int64))
    q = db[5 * 2320 + 100:5 * 2320 + 405].copy()
    off = np.array([0, q.size], np.int64)
    for name, fn in (("find (top-1)", lambda: g.search_topk(q, off, 1)), ("find_votes", lambda: g.search_votes(q, off))):
        fn()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        print(f"{n_clips} clips, {name}: p50 {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f} ms")
