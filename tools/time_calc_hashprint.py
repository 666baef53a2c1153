"""Latency of the one-file entry point par_collector_calc_hashprint on a 5 s WAV (python3 tools/time_calc_hashprint.py)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

d = tempfile.mkdtemp(prefix="hpfw_one_")
p = os.path.join(d, "q.wav")
synth.write_wav(p, synth.gen_clip(3, 5.0))
cache = os.path.join(d, "cache") + "/"
os.makedirs(cache)
with open(cache + "filters.cereal", "wb") as f:
    f.write(np.array([64, 2420], np.int32).tobytes() + synth.make_filters().astype(np.float32).tobytes())
pc = hpfw_amd.ParallelCollector()
pc.load(cache)
pc.calc_hashprint(p)
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    pc.calc_hashprint(p)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"calc_hashprint(5 s file): p50 {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f} ms")
