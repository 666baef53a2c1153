"""Phases of hamming_shift_kernel per workgroup (library built with -DHPFW_SHIFT_STAMPS; run on the GPU box)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpfw_amd
n_clips, kq, n_hp = 125000, 304, 2320
gpu = hpfw_amd.Gpu(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_clips, n_hp), dtype=torch.int64, generator=g, device=dev)
stream = torch.cuda.current_stream().cuda_stream
gpu.index_clear()
gpu.index_add_dev(db.data_ptr(), np.arange(0, (n_clips + 1) * n_hp, n_hp, dtype=np.int64), stream)
hits = torch.empty((1, 5, 4), dtype=torch.int32, device=dev)
q = db[77, 100:100 + kq].clone()
for _ in range(3):
    gpu.search_topk_dev(q.data_ptr(), np.array([0, kq], dtype=np.int64), 5, hits.data_ptr(), stream)
torch.cuda.synchronize()
L = hpfw_amd.lib(); L.hpfw_debug_shift_stamps.restype = ctypes.c_void_p
ptr = L.hpfw_debug_shift_stamps()
nwg = 512
buf = (ctypes.c_longlong * (nwg * 8))()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, ctypes.c_void_p(ptr), nwg * 64, 2)
a = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 8)
names = ["first window", "matrix loop (+ next window's loads issued)", "barrier", "partial sums + minimum", "barrier", "next window: wait, expand, write", "barrier"]
tot = a[:, :7].sum(axis=1)
print("ticks of s_memtime (100 MHz): median total per workgroup", np.median(tot), " first start", a[:, 7].min(), " last start", a[:, 7].max())
for k, nm in enumerate(names):
    print(f"{nm:32s} median {np.median(a[:, k]):8.0f}  mean {a[:, k].mean():8.1f}  share {a[:, k].sum() / tot.sum():.3f}")
