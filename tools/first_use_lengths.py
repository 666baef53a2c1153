"""A corpus of real recordings: every file has a length of its own, so every file pays for the tables of its length
once.  Times the first extraction of clips of distinct lengths (plan + tables + one clip) and the steady state of one
of them.  python tools/first_use_lengths.py [seconds] [files]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
files = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
rng = np.random.default_rng(5)
base = int(sec * 44100)
lengths = [base + int(d) for d in rng.choice(20000, files, replace=False)]


def seven_smooth(n):
    for f in (2, 3, 5, 7):
        while n % f == 0:
            n //= f
    return n == 1


smooth = sum(1 for n in lengths if seven_smooth(n))
pcm = torch.randint(-3000, 3000, (max(lengths),), dtype=torch.int16, device="cuda")
hp = torch.empty((g.geometry(max(lengths)).n_hp + 8,), dtype=torch.int64, device="cuda")
g.extract_dev(pcm.data_ptr(), base * 2 // 2 - 1, 1, hp.data_ptr())          # warm the kernels up on another length
torch.cuda.synchronize()
t0 = time.perf_counter()
per = []
for n in lengths:
    t1 = time.perf_counter()
    g.extract_dev(pcm.data_ptr(), n, 1, hp.data_ptr())
    torch.cuda.synchronize()
    per.append(time.perf_counter() - t1)
dt = time.perf_counter() - t0
print(f"{files} files of distinct lengths around {sec:.0f} s ({smooth} of them 7-smooth): {files / dt:7.1f} files/s, "
      f"first use {1e3 * np.median(per):.1f} ms median, {1e3 * max(per):.1f} ms worst", flush=True)
t0 = time.perf_counter()
for n in lengths:
    g.extract_dev(pcm.data_ptr(), n, 1, hp.data_ptr())
torch.cuda.synchronize()
print(f"second pass over the same lengths (plans cached while they fit): {files / (time.perf_counter() - t0):7.1f} files/s", flush=True)
