"""Does the whole extraction overlap with itself across streams?  k handles on k streams, 1000 / k clips each,
against one handle on all 1000 clips.  python tools/overlap_extract.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth

n, total = 1323000, 1000
filt = synth.make_filters()
pcm = torch.randint(-3000, 3000, (total, n), dtype=torch.int16, device="cuda")
for k in (1, 2, 4):
    hs = [hpfw_amd.Gpu(0) for _ in range(k)]
    for h in hs:
        h.set_filters(filt)
    geo = hs[0].geometry(n)
    hp = torch.empty((total, geo.n_hp), dtype=torch.int64, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(k)]
    per = total // k

    def go():
        for i, (h, s) in enumerate(zip(hs, streams)):
            h.extract_dev(pcm[i * per].data_ptr(), n, per, hp[i * per].data_ptr(), s.cuda_stream)

    go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    print(f"{k} stream(s) x {per} clips: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 1000 clips", flush=True)
    del hs
