"""Steady state of the chirp-z forward path at one length, for rocprofv3 --kernel-trace --stats.
python tools/chirpz_profile.py [n_samples] [clips] [reps]"""
import sys

import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1323001
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
geo = g.geometry(n)
pcm = torch.randint(-3000, 3000, (n_clips, n), dtype=torch.int16, device="cuda")
hp = torch.empty((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
for _ in range(reps):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
print("done", n, geo.n1, geo.n2)
