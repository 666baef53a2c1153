import os, sys, ctypes
os.environ["HPFW_COLS_STAMPS"] = "1"
os.environ["HPFW_FWD_CHUNK"] = "0"   # one launch over the whole batch: the stamps are indexed by the launch's workgroup id
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import hpfw_amd
from hpfw_amd import synth
n_clips, n = 1000, 1323000
g = hpfw_amd.Gpu(0); g.set_filters(synth.make_filters()); g.set_batch(n_clips)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
hp = torch.zeros(n_clips, g.geometry(n).n_hp, dtype=torch.int64, device="cuda")
for _ in range(2):
    g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
L = hpfw_amd.lib(); L.hpfw_debug_cols_stamps.restype = ctypes.c_void_p
ptr = L.hpfw_debug_cols_stamps()
nwg = n_clips * 50
buf = (ctypes.c_longlong * (nwg * 8))()
import ctypes.util
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, ctypes.c_void_p(ptr), nwg * 64, 2)
a = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 8)
names = ["digits issue + samples into registers", "wait + barrier", "matrix loop", "wait for corr", "next digits issue", "conversion", "stores issue", "-"]
tot = a.sum(axis=1)
a = a[tot > 0]; tot = tot[tot > 0]
print("workgroups", a.shape[0], "; cycles per workgroup (wave 0, s_memtime): median total", np.median(tot))
for k, nm in enumerate(names):
    print(f"{nm:36s} median {np.median(a[:, k]):10.0f}  mean {a[:, k].mean():10.0f}  share {a[:, k].sum() / tot.sum():.3f}")
