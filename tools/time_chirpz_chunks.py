"""Any-length (chirp-z) extraction rate and a checksum of the hashprints, for A/B runs of HPFW_BZ_CHUNK / HPFW_FWD_STREAMS:
python tools/time_chirpz_chunks.py [n_samples] [clips] [reps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1323001
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
geo = g.geometry(n)
gen = torch.Generator(device="cuda").manual_seed(7)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.empty((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
for _ in range(2):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{n_clips / dt:9.0f} clips/s  {dt * 1e3:7.3f} ms per pass  checksum {int(hp.sum().item()) & 0xffffffffffff:012x}")
