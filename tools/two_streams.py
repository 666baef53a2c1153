"""Experiment: does running two halves of the batch on two streams (two handles) overlap the MFMA-bound
back end of one with the latency-bound front end of the other?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1323000
filt = synth.make_filters()
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.zeros(n_clips, 2320, dtype=torch.int64, device="cuda")
g0 = hpfw_amd.Gpu(0)
g0.set_filters(filt)


def single():
    g0.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())


single()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    single()
torch.cuda.synchronize()
print(f"one stream: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
ref = hp.clone()
hp.zero_()
hs = [hpfw_amd.Gpu(0) for _ in range(2)]
ss = [torch.cuda.Stream() for _ in range(2)]
for h in hs:
    h.set_filters(filt)
per = n_clips // parts


def dual():
    for p in range(parts):
        h, s = hs[p % 2], ss[p % 2]
        lo = p * per
        cnt = per if p < parts - 1 else n_clips - lo
        h.extract_dev(pcm[lo].data_ptr(), n, cnt, hp[lo].data_ptr(), s.cuda_stream)


dual()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    dual()
torch.cuda.synchronize()
print(f"two streams, {parts} parts: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms; identical: {bool(torch.equal(hp, ref))}")
