"""A few extraction passes over synthetic clips resident in HBM -- the profiling workload
(rocprofv3 --kernel-trace / --pmc ... -- python3 tools/extract_pass.py [clips] [passes])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.zeros(n_clips, g.geometry(n).n_hp, dtype=torch.int64, device="cuda")
for _ in range(passes):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
print("ok", int(hp[0, 0].item()) & 0xffff)
