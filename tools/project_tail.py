"""Is project_kernel's time quantised by rounds of resident workgroups?  10 workgroups per 30 s clip, 3 per CU x 256 CUs
= 768 resident: 998 clips = 12.99 rounds, 1000 = 13.02, 1075 = 14.0.  python tools/project_tail.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
c = 2419
s = (torch.rand((1100, 121, c), device="cuda") * -80.0).contiguous()
proj = torch.empty((1100, 64, c - 19), dtype=torch.float32, device="cuda")
for n in (921, 998, 1000, 1024, 1075, 1076):
    for _ in range(2):
        g.stage_project_dev(s.data_ptr(), n, c, proj.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.stage_project_dev(s.data_ptr(), n, c, proj.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{n:5d} clips ({n * 10 / 768:6.2f} rounds): {dt:6.3f} ms  {dt / n * 1e3:6.3f} us per clip  "
          f"{2 * 64 * 2420 * 2400 * n / dt / 1e9:6.1f} TFLOP/s", flush=True)
