"""Throughput of the file entry point par_collector_prepare (read WAV -> covariance -> filters ->
hashprints) on synthetic 30 s files in a scratch directory (python3 tools/time_prepare.py [files])."""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 400
d = tempfile.mkdtemp(prefix="hpfw_prepare_")
try:
    rng = np.random.default_rng(0)
    paths = []
    for i in range(n_files):
        p = os.path.join(d, f"f{i:05d}.wav")
        synth.write_wav(p, rng.integers(-3000, 3000, 1323000, dtype=np.int16))
        paths.append(p)
    pc = hpfw_amd.ParallelCollector()
    pc.load(os.path.join(d, "cache") + "/")
    for label in ("cold handle", "warm"):
        t0 = time.perf_counter()
        res = pc.prepare(paths)
        dt = time.perf_counter() - t0
        print(f"prepare ({label}): {len(res)} of {n_files} files in {dt:.2f} s = {len(res) / dt:.0f} files/s "
              f"({n_files * 2.646 / dt / 1e3:.2f} GB/s of WAV)")
    os.environ["HPFW_PREPARE_KEEP_FILTERS"] = "1"
    t0 = time.perf_counter()
    res = pc.prepare(paths)
    dt = time.perf_counter() - t0
    print(f"prepare (filters kept, no learning): {len(res) / dt:.0f} files/s")
finally:
    shutil.rmtree(d, ignore_errors=True)
