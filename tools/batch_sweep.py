"""Front-end pass size sweep: does keeping a sub-batch's intermediates (Y' = 5.3 MB per 30 s clip) inside the 256 MiB
Infinity Cache pay for the launch tails of more, smaller launches?  1000 x 30 s clips, hpfw_gpu_set_batch(b).
python tools/batch_sweep.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n_clips, n = 1000, 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
pcm = torch.randint(-3000, 3000, (n_clips, n), dtype=torch.int16, device="cuda")
hp = torch.empty((n_clips, g.geometry(n).n_hp), dtype=torch.int64, device="cuda")
for b in (1024, 500, 250, 128, 64, 40, 20):
    g.set_batch(b)
    for _ in range(2):
        g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize()
    g.set_kernel_timing(-1)
    t0 = time.perf_counter()
    for _ in range(3):
        g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    kt = {k: round(v[0] / 3, 2) for k, v in g.kernel_timing().items() if v[1]}
    g.set_kernel_timing(0)
    print(f"batch {b:5d}: {dt:6.2f} ms per 1000 clips  {kt}", flush=True)
