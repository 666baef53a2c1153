"""Throughput of the file entry point par_collector_prepare (read WAV -> covariance -> filters ->
hashprints) on synthetic 30 s files in a scratch directory (python3 tools/time_prepare.py [files]);
with a second argument "varied" the files have random lengths between 60 and 240 s, as full tracks do --
then almost every file brings a new clip length and its tables."""
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
varied = len(sys.argv) > 2 and sys.argv[2] == "varied"
d = tempfile.mkdtemp(prefix="hpfw_prepare_")
try:
    rng = np.random.default_rng(0)
    paths = []
    for i in range(n_files):
        p = os.path.join(d, f"f{i:05d}.wav")
        n = int(rng.integers(60 * 44100, 240 * 44100)) if varied else 1323000
        synth.write_wav(p, rng.integers(-3000, 3000, n, dtype=np.int16))
        paths.append(p)
    pc = hpfw_amd.ParallelCollector()
    pc.load(os.path.join(d, "cache") + "/")
    for label in ("cold handle", "warm"):
        t0 = time.perf_counter()
        res = pc.prepare(paths)
        dt = time.perf_counter() - t0
        gb = sum(os.path.getsize(q) for q in paths) / 1e9
        print(f"prepare ({label}): {len(res)} of {n_files} files in {dt:.2f} s = {len(res) / dt:.1f} files/s "
              f"({gb / dt:.2f} GB/s of WAV)")
    os.environ["HPFW_PREPARE_KEEP_FILTERS"] = "1"
    t0 = time.perf_counter()
    res = pc.prepare(paths)
    dt = time.perf_counter() - t0
    print(f"prepare (filters kept, no learning): {len(res) / dt:.1f} files/s")
finally:
    shutil.rmtree(d, ignore_errors=True)
