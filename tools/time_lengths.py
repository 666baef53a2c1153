"""Extraction rate of clip lengths around 30 s: 7-smooth (mixed-radix forward transform) against lengths with a
prime factor above 7 (chirp-z forward transform), inputs resident in HBM.  python tools/time_lengths.py [clips]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
g.set_kernel_timing(-1)
for n in (1323000, 1323001, 1322999, 220500, 220501):
    geo = g.geometry(n)
    pcm = torch.randint(-3000, 3000, (n_clips, n), dtype=torch.int16, device="cuda")
    hp = torch.empty((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
    for _ in range(2):
        g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize()
    g.set_kernel_timing(-1)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kt = {k: round(v[0] / reps, 3) for k, v in g.kernel_timing().items() if v[1]}
    print(f"N={n} n1={geo.n1} n2={geo.n2}: {n_clips / dt:9.0f} clips/s  {dt * 1e3:7.2f} ms per {n_clips} clips  {kt}", flush=True)
    del pcm, hp
