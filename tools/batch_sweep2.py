import sys, time, torch
sys.path.insert(0, ".")
import hpfw_amd
from hpfw_amd import synth
n, n_clips = 1323000, 1000
g = hpfw_amd.Gpu(0); g.set_filters(synth.make_filters())
geo = g.geometry(n)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.empty((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
for b in [int(a) for a in sys.argv[1:]]:
    g.set_batch(b)
    for _ in range(2): g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"batch {b:5d}: {dt*1e3:7.3f} ms per 1000 clips  checksum {int(hp.sum().item()) & 0xffffffffffff:012x}")
