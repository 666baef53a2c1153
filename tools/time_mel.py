"""Throughput of the Mel front-end on clips resident in HBM (python3 tools/time_mel.py [clips])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hpfw_amd  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 1323000
g = hpfw_amd.Gpu(0)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
nf = int(hpfw_amd.lib().hpfw_gpu_mel_frames(n))
out = torch.zeros(n_clips, 33, nf, device="cuda")
cols = torch.zeros(n_clips, dtype=torch.int32, device="cuda")
L = hpfw_amd.lib()
for _ in range(2):
    hpfw_amd._lib.check(L.hpfw_gpu_mel_spectrogram_pcm16(g._h, pcm.data_ptr(), n, n_clips, out.data_ptr(), cols.data_ptr(), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
hpfw_amd._lib.check(L.hpfw_gpu_mel_spectrogram_pcm16(g._h, pcm.data_ptr(), n, n_clips, out.data_ptr(), cols.data_ptr(), None))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"mel: {n_clips} x 30 s clips in {dt * 1e3:.1f} ms = {n_clips / dt:.0f} clips/s, {nf} frames per clip, kept {int(cols[0])}")
