"""rows_snapshots.py -- the row transform beside hashprint_q_kernel, with the LDS image of every workgroup kept after
the load and after each fused group (build: make OUT=../lib_snap EXTRA=-DHPFW_ROWS_SNAP): which elements go wrong
first, and what do they hold?   HPFW_GPU_LIB=hpfw_amd/lib_snap/libhpfw_gpu.so python tools/rows_snapshots.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

NC = 32
filt = synth.make_filters()
base = np.stack([synth.gen_clip(4000 + i, 30.0) for i in range(4)])
clips = np.concatenate([np.roll(base, 53 * r, axis=1) for r in range(NC // 4)])
n = clips.shape[1]
plan = oracle.Plan(n)
nk = plan.kmax - plan.kmin
n1, n2 = plan.n1, plan.n2
hq = n1 // 2 + 1
d = torch.from_numpy(clips).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
os.environ["HPFW_FWD_CHUNK"] = "0"
os.environ["HPFW_CQ_SERIAL"] = "1"
vic, agg = hpfw_amd.Gpu(0), hpfw_amd.Gpu(0)
for g in (vic, agg):
    g.set_filters(filt)
L = hpfw_amd.lib()
L.hpfw_gpu_debug_set_rows_snap.argtypes = [ctypes.c_void_p]
snap = torch.zeros((hq, NC, 4, n2, 2), dtype=torch.float32, device="cuda")   # workgroup = blockIdx.y (q1) * gridDim.x (clips) + clip
L.hpfw_gpu_debug_set_rows_snap(snap.data_ptr())
x_ref = torch.zeros((NC, nk, 2), dtype=torch.float32, device="cuda")
vic.stage_spectrum_dev(d.data_ptr(), n, NC, x_ref.data_ptr())
torch.cuda.synchronize()
want = plan.spectrum(clips[0])
assert np.array_equal(x_ref[0].cpu().numpy().view(np.uint32), want.view(np.uint32))
snap_ref = snap.clone()
mag = torch.zeros((NC, 121, plan.c), dtype=torch.float32, device="cuda")
vic.stage_cqmag_dev(x_ref.data_ptr(), n, NC, mag.data_ptr())
db = torch.zeros_like(mag)
vic.stage_db_dev(mag.data_ptr(), NC, plan.c, db.data_ptr())
hp = torch.zeros((NC, plan.n_hp), dtype=torch.int64, device="cuda")
out = torch.zeros_like(x_ref)
torch.cuda.synchronize()
# stability of the snapshots themselves
vic.stage_spectrum_dev(d.data_ptr(), n, NC, out.data_ptr())
torch.cuda.synchronize()
assert bool((snap.view(torch.int32) == snap_ref.view(torch.int32)).all()), "snapshots differ without an aggressor"
shown = 0
for rnd in range(20):
    snap.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        agg.hashprints_from_db_dev(db.data_ptr(), NC, plan.c, hp.data_ptr(), sb.cuda_stream)
    vic.stage_spectrum_dev(d.data_ptr(), n, NC, out.data_ptr(), sa.cuda_stream)
    for _ in range(3):
        agg.hashprints_from_db_dev(db.data_ptr(), NC, plan.c, hp.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    xbad = int((out.view(torch.int32) != x_ref.view(torch.int32)).any(dim=2).any(dim=1).sum())
    ne = snap.view(torch.int32) != snap_ref.view(torch.int32)           # [hq][NC][4][n2][2]
    per = ne.any(dim=4).sum(dim=3)                                     # elements differing per (q1, clip, slot)
    wgs = torch.nonzero(per.sum(dim=2))
    print(f"round {rnd}: {xbad} clips with a wrong spectrum, {wgs.shape[0]} workgroups with a differing snapshot", flush=True)
    for q1, c in wgs.tolist()[:6 if shown < 40 else 0]:
        counts = per[q1, c].tolist()
        first = next(s for s in range(4) if counts[s])
        idx = torch.nonzero(ne[q1, c, first].any(dim=1)).flatten()
        got = snap[q1, c, first][idx].cpu().numpy()
        ref = snap_ref[q1, c, first][idx].cpu().numpy()
        prev = snap_ref[q1, c, first - 1][idx].cpu().numpy() if first else None
        i = idx.cpu().numpy()
        print(f"  q1 {q1} clip {c}: differing elements per snapshot {counts}; first bad snapshot {first}: {i.size} elements, indices {i[:24].tolist()}")
        for j in range(min(4, i.size)):
            line = f"     [{i[j]}] got ({got[j][0]:.6g}, {got[j][1]:.6g}) want ({ref[j][0]:.6g}, {ref[j][1]:.6g})"
            if prev is not None:
                line += f" before the group ({prev[j][0]:.6g}, {prev[j][1]:.6g})"
            print(line)
        shown += 1
