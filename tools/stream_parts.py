"""Where the time of one streaming query goes: extraction of one 5 s window, the scan, the top-k."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
gen = torch.Generator(device="cuda").manual_seed(3)
db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_clips, 2320), dtype=torch.int64, generator=gen, device="cuda")
g.index_add_dev(db.data_ptr(), np.arange(n_clips + 1, dtype=np.int64) * 2320)
n = 220500
pcm = (torch.randn(1, n, device="cuda", generator=gen) * 3000).to(torch.int16)
geo = g.geometry(n)
hp = torch.zeros((1, geo.n_hp), dtype=torch.int64, device="cuda")
hits = torch.zeros((1, 10, 4), dtype=torch.int32, device="cuda")
q_off = np.array([0, geo.n_hp], np.int64)


def timed(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"extract one 5 s window: {timed(lambda: g.extract_dev(pcm.data_ptr(), n, 1, hp.data_ptr())):.3f} ms")
print(f"search (scan + top-k):  {timed(lambda: g.search_topk_dev(hp.data_ptr(), q_off, 10, hits.data_ptr())):.3f} ms")
g.set_kernel_timing(-1)
g.search_topk_dev(hp.data_ptr(), q_off, 10, hits.data_ptr())
torch.cuda.synchronize()
print({k: round(v[0], 3) for k, v in g.kernel_timing().items() if v[1]})
print(f"hits to host: {timed(lambda: hits.cpu()):.3f} ms")
