"""q_stamps.py -- where a workgroup of hashprint_q_kernel spends its time (build: make OUT=../lib_qst EXTRA=-DHPFW_Q_STAMPS):
s_memtime ticks (= shader cycles) of wave 0 at the phase boundaries, over the workgroups of one launch (dB spectrograms in).
    HPFW_GPU_LIB=hpfw_amd/lib_qst/libhpfw_gpu.so python tools/q_stamps.py [clips] [extract]
"extract": the launches of a whole extraction (dB terms fresh from the chirp-z stage) instead of the dB-input entry point on
random values from HBM"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 500
EXTRACT = len(sys.argv) > 2 and sys.argv[2] == "extract"   # the extraction's own launches (dB terms fresh from the chirp-z stage)
c = 2419
nhp = c - 99
tiles = (nhp + 127) // 128
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
db = (-80.0 * torch.rand((n_clips, 121, c), device="cuda")).contiguous()
hp = torch.zeros((n_clips, nhp), dtype=torch.int64, device="cuda")
st = torch.zeros((tiles * n_clips + 8, 8), dtype=torch.int64, device="cuda")
if EXTRACT:
    import ctypes
    n = 1323000
    g.set_batch(n_clips)                                   # one pass: workgroup ids = (clip, tile) of the whole batch
    gen = torch.Generator(device="cuda").manual_seed(1)
    pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
    L = hpfw_amd.lib()
    L.hpfw_gpu_debug_set_q_stamps.argtypes = [ctypes.c_void_p]
    L.hpfw_gpu_debug_set_q_stamps(st.data_ptr())
    for _ in range(3):
        g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
else:
    for _ in range(3):
        g.stage_delta_q_dev(db.data_ptr(), n_clips, c, st.data_ptr(), hp.data_ptr())
torch.cuda.synchronize()
a = st.cpu().numpy()[: tiles * n_clips]
d = np.diff(a[:, :6], axis=1).astype(np.float64)
tot = (a[:, 5] - a[:, 0]).astype(np.float64)
names = ["staging (loads, quantisation, LDS writes)", "barrier", "main loop (40 steps of 72 matrix instructions)", "epilogue (int64 sums, signs, shuffles)",
         "barrier + store"]
print(f"workgroups {a.shape[0]}; cycles per workgroup: median {np.median(tot):.0f}, mean {tot.mean():.0f}")
for k, nm in enumerate(names):
    print(f"{nm:52s} median {np.median(d[:, k]):7.0f}  mean {d[:, k].mean():8.1f}  share {d[:, k].sum() / tot.sum():.3f}")
# how far apart do the two workgroups of a CU run?  for each workgroup: the share of its main loop during which no other
# workgroup ON THE SAME CU (HW_ID bits 8..15: CU, SH, SE; XCC from the dispatch order is not in HW_ID: id mod 8) is in its main loop
hw = a[:, 6]
cu = ((hw >> 8) & 0xFF) | ((a[:, 7] & 7) << 8)
alone = []
for key in np.unique(cu)[:64]:
    idx = np.nonzero(cu == key)[0]
    iv = sorted((a[i, 2], a[i, 3]) for i in idx)
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last, t1, t2 = 0, ev[0][0], 0, 0
    for t, dlt in ev:
        if depth == 1:
            t1 += t - last
        elif depth >= 2:
            t2 += t - last
        depth += dlt
        last = t
    alone.append((t1, t2, iv[-1][1] - iv[0][0]))
alone = np.array(alone, np.float64)
print(f"per CU (first 64): main loops cover {((alone[:, 0] + alone[:, 1]) / alone[:, 2]).mean():.3f} of the time; two at once {(alone[:, 1] / alone[:, 2]).mean():.3f}, one alone {(alone[:, 0] / alone[:, 2]).mean():.3f}")
