"""Do two stages of the extraction path overlap when they run on two streams (two handles)?  Times each stage
alone and every pair together over the same 1000 x 30 s clips: rows+cols (stage_spectrum), chirp-z (stage_cqmag),
projection (stage_project).  python tools/overlap_probe.py [clips]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 1323000
ga, gb = hpfw_amd.Gpu(0), hpfw_amd.Gpu(0)
filt = synth.make_filters()
ga.set_filters(filt), gb.set_filters(filt)
geo = ga.geometry(n)
nk = geo.kmax - geo.kmin
pcm = torch.randint(-3000, 3000, (n_clips, n), dtype=torch.int16, device="cuda")
x = torch.empty((n_clips, nk, 2), dtype=torch.float32, device="cuda")
x2 = torch.empty_like(x)
mag = torch.empty((n_clips, 121, geo.c), dtype=torch.float32, device="cuda")
db = torch.empty_like(mag)
proj = torch.empty((n_clips, 64, geo.n_frames), dtype=torch.float32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ga.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, x.data_ptr())
ga.stage_cqmag_dev(x.data_ptr(), n, n_clips, mag.data_ptr())
ga.stage_db_dev(mag.data_ptr(), n_clips, geo.c, db.data_ptr())
gb.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, x2.data_ptr())
gb.stage_cqmag_dev(x.data_ptr(), n, n_clips, mag.data_ptr())
gb.stage_project_dev(db.data_ptr(), n_clips, geo.c, proj.data_ptr())
torch.cuda.synchronize()
stages = {
    "spectrum": lambda g, s, out: g.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, out.data_ptr(), s.cuda_stream),
    "cqmag": lambda g, s, out: g.stage_cqmag_dev(x.data_ptr(), n, n_clips, mag.data_ptr(), s.cuda_stream),
    "project": lambda g, s, out: g.stage_project_dev(db.data_ptr(), n_clips, geo.c, proj.data_ptr(), s.cuda_stream),
}


def run(pairs, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for name, g, s, out in pairs:
            stages[name](g, s, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


alone = {k: run([(k, ga, s1, x)]) for k in stages}
print("alone (ms):", {k: round(v, 2) for k, v in alone.items()}, flush=True)
for a, b in (("spectrum", "project"), ("cqmag", "project"), ("spectrum", "cqmag")):
    both = run([(a, ga, s1, x), (b, gb, s2, x2)])
    print(f"{a} || {b}: {both:.2f} ms  (sum {alone[a] + alone[b]:.2f}, max {max(alone[a], alone[b]):.2f})", flush=True)
