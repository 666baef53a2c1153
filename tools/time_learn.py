"""Times the index()-only legs on one GPU: covariance accumulation and the filter eigen-solve."""
import sys
import time

import numpy as np
import torch

import hpfw_amd

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 1323000
g = hpfw_amd.Gpu(0)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
g.cov_accumulate_dev(pcm.data_ptr(), n, n_clips)
torch.cuda.synchronize()
g.cov_reset()
g.set_kernel_timing(1)
t0 = time.perf_counter()
g.cov_accumulate_dev(pcm.data_ptr(), n, n_clips)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"cov_accumulate {n_clips} clips: {dt*1e3:.1f} ms = {n_clips/dt:.0f} clips/s")
try:
    print(g.kernel_timing())
except Exception as e:  # noqa
    print("timing:", e)
t0 = time.perf_counter()
f = g.learn_filters()
print(f"learn_filters: {time.perf_counter()-t0:.2f} s", f.shape)
