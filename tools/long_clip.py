"""Extraction of long clips: parity against the oracle for one clip and throughput for a small batch
(python3 tools/long_clip.py [seconds] [clips])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(round(seconds * 44100))
filt = synth.make_filters()
g = hpfw_amd.Gpu(0)
g.set_filters(filt)
geo = g.geometry(n)
print(f"{seconds:g} s: n1 {geo.n1} n2 {geo.n2} C {geo.c} hashprints {geo.n_hp}")
clip = synth.gen_clip(3, seconds)
t0 = time.perf_counter()
got = g.extract(clip[None, :])[0]
print(f"first extract (plan + tables): {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter()
want = oracle.Plan(n).extract(filt, clip)
print(f"oracle: {time.perf_counter() - t0:.1f} s; bit-identical: {bool(np.array_equal(got, want))}")
pcm = torch.from_numpy(np.stack([clip] * n_clips)).cuda()
hp = torch.zeros((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
g.set_kernel_timing(-1)
t0 = time.perf_counter()
g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n_clips} clips: {dt * 1e3:.1f} ms = {n_clips * seconds / dt:.0f} x real time; kernels {g.kernel_timing()}")
