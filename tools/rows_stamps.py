"""rows_stamps.py -- where a workgroup of the row transform spends its time (build: make OUT=../lib_stamps
EXTRA=-DHPFW_ROWS_STAMPS): s_memtime ticks (100 MHz) of wave 0 up to each barrier, averaged over the workgroups of one
launch over 1000 clips.   HPFW_GPU_LIB=hpfw_amd/lib_stamps/libhpfw_gpu.so python tools/rows_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HPFW_FWD_CHUNK"] = "0"
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips, n = 500, 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
g.set_batch(n_clips)
geo = g.geometry(n)
hq = geo.n1 // 2 + 1
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
nk = geo.kmax - geo.kmin
x = torch.zeros((n_clips, nk, 2), dtype=torch.float32, device="cuda")
L = hpfw_amd.lib()
L.hpfw_gpu_debug_set_rows_snap.argtypes = [ctypes.c_void_p]
st = torch.zeros((hq * n_clips, 8), dtype=torch.int64, device="cuda")
L.hpfw_gpu_debug_set_rows_snap(st.data_ptr())
for _ in range(3):
    g.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, x.data_ptr())
torch.cuda.synchronize()
a = st.cpu().numpy()
d = np.diff(a, axis=1)
names = ["z loads, twiddle seeds, LDS writes, barrier", "group (7, 3) in place, barrier", "group (5, 3) compute, barrier", "group (5, 3) transposed stores, barrier",
         "group (5, 4) compute, barrier", "group (5, 4) stores, barrier", "pruned output stores"]
tot = a[:, 7] - a[:, 0]
print(f"workgroups {a.shape[0]}; ticks per workgroup (100 MHz): median {np.median(tot):.0f} = {np.median(tot) / 100:.1f} us, mean {tot.mean():.0f}")
for k, nm in enumerate(names):
    print(f"{nm:48s} median {np.median(d[:, k]):7.0f}  mean {d[:, k].mean():8.1f}  share {d[:, k].sum() / tot.sum():.3f}")
span = (a[:, 7].max() - a[:, 0].min()) / 100.0
print(f"first start to last end: {span:.0f} us; workgroup-time / span = {tot.sum() / 100.0 / span:.1f} workgroups in flight ({tot.sum() / 100.0 / span / 256:.2f} per CU)")
