"""rows_error_shape.py -- the forward transform beside hashprint_q_kernel: keep the wrong spectra for offline analysis."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

NC = 48
filt = synth.make_filters()
base = np.stack([synth.gen_clip(4000 + i, 30.0) for i in range(6)])
clips = np.concatenate([np.roll(base, 53 * r, axis=1) for r in range(NC // 6)])
n = clips.shape[1]
plan = oracle.Plan(n)
nk = plan.kmax - plan.kmin
d = torch.from_numpy(clips).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
os.environ["HPFW_FWD_CHUNK"] = "0"
os.environ["HPFW_CQ_SERIAL"] = "1"
vic, agg = hpfw_amd.Gpu(0), hpfw_amd.Gpu(0)
for g in (vic, agg):
    g.set_filters(filt)
x_ref = torch.zeros((NC, nk, 2), dtype=torch.float32, device="cuda")
vic.stage_spectrum_dev(d.data_ptr(), n, NC, x_ref.data_ptr())
mag = torch.zeros((NC, 121, plan.c), dtype=torch.float32, device="cuda")
vic.stage_cqmag_dev(x_ref.data_ptr(), n, NC, mag.data_ptr())
db = torch.zeros_like(mag)
vic.stage_db_dev(mag.data_ptr(), NC, plan.c, db.data_ptr())
hp = torch.zeros((NC, plan.n_hp), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
zp, _ = vic.debug_workspace(0)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hq = plan.n1 // 2 + 1
def fetch_z():
    out = np.empty((NC, (plan.n2 + 127) // 128, 2 * hq, 128), np.float32)   # [column block][row][128]: kernels.h kZBlock
    assert hip.hipMemcpy(out.ctypes.data, zp, out.nbytes, 2) == 0
    return out
z_ref = fetch_z()
xr = x_ref.cpu().numpy()
out = torch.zeros_like(x_ref)
kept = {}
z_bad_total = 0
for rnd in range(30):
    agg.hashprints_from_db_dev(db.data_ptr(), NC, plan.c, hp.data_ptr(), sb.cuda_stream)
    vic.stage_spectrum_dev(d.data_ptr(), n, NC, out.data_ptr(), sa.cuda_stream)
    agg.hashprints_from_db_dev(db.data_ptr(), NC, plan.c, hp.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    x = out.cpu().numpy()
    z = fetch_z()
    zne = (z.view(np.uint32) != z_ref.view(np.uint32))
    z_bad_total += int(zne.sum())
    ne = (x.view(np.uint32) != xr.view(np.uint32)).any(axis=2)
    for c in np.nonzero(ne.any(axis=1))[0]:
        rows = np.unique((np.nonzero(ne[c])[0] + plan.kmin) % plan.n1)
        zrows = np.nonzero(zne[c].any(axis=(0, 2)))[0]          # rows 2 q1 + (Re: 0, Im: 1)
        print(f"round {rnd} clip {c}: {int(ne[c].sum())} bins, rows {rows.tolist()[:10]}, z rows differing {zrows.tolist()[:10]}", flush=True)
        if len(kept) < 14:
            kept[f"bad_{rnd}_{c}"] = x[c].copy()
            kept[f"ref_{rnd}_{c}"] = xr[c].copy()
print("z values differing in total:", z_bad_total)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "diag5", "rows_errors.npz"), kmin=plan.kmin, n1=plan.n1, n2=plan.n2, **kept)
