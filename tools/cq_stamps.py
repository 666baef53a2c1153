"""cq_stamps.py -- where a workgroup (one band of one clip) of the chirp-z stage spends its time (build: make OUT=../lib_cqst
EXTRA=-DHPFW_CQ_STAMPS): s_memtime (= shader cycles) of thread 0 at the phase boundaries, per size class, over the launches
of a whole extraction.   HPFW_GPU_LIB=hpfw_amd/lib_cqst/libhpfw_gpu.so python tools/cq_stamps.py [clips]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
g.set_batch(n_clips)
geo = g.geometry(n)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.zeros((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
st = torch.zeros((n_clips, 121, 8), dtype=torch.int64, device="cuda")
L = hpfw_amd.lib()
L.hpfw_gpu_debug_set_cq_stamps.argtypes = [ctypes.c_void_p]
L.hpfw_gpu_debug_set_cq_stamps(st.data_ptr())
for _ in range(3):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 8)
names = ["load: X[bins] * window -> LDS (+ zero padding)", "barrier", "both transforms and the product between them", "magnitudes, dB terms, stores", "wave maxima"]
for p in sorted(set(a[:, 6].tolist())):
    b = a[a[:, 6] == p]
    d = np.diff(b[:, :6], axis=1).astype(np.float64)
    tot = (b[:, 5] - b[:, 0]).astype(np.float64)
    print(f"class {int(p):6d}: {b.shape[0]} workgroups, cycles per workgroup median {np.median(tot):.0f} mean {tot.mean():.0f}")
    for k, nm in enumerate(names):
        print(f"    {nm:50s} median {np.median(d[:, k]):7.0f}  share {d[:, k].sum() / tot.sum():.3f}")
