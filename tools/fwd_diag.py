"""fwd_diag.py -- the forward transform of 128 x 30 s clips under switches, z (column stage) and x (row stage) against a
reference run (one launch per stage, one stream, default kernel) that itself equals the oracle.  One process.

  python tools/fwd_diag.py [reps] [seconds] [n_clips]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
n_clips = int(sys.argv[3]) if len(sys.argv) > 3 else 128
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def fetch(ptr, count, dtype=np.float32):
    out = np.empty(count, dtype)
    rc = hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2)
    assert rc == 0, rc
    return out


def handle(env):
    for k, v in env.items():
        os.environ[k] = v
    g = hpfw_amd.Gpu(0)
    for k in env:
        del os.environ[k]
    return g


n_base = max(1, n_clips // 10)
base = np.stack([synth.gen_clip(4000 + i, seconds) for i in range(n_base)])
clips = np.concatenate([np.roll(base, 53 * r, axis=1) for r in range((n_clips + n_base - 1) // n_base)])[:n_clips]
n = clips.shape[1]
plan = oracle.Plan(n)
nk = plan.kmax - plan.kmin
n1, n2 = plan.n1, plan.n2
hq = n1 // 2 + 1
zclip = (n2 + 127) // 128 * 2 * hq * 128   # floats of z per clip: [column block][row][128] (kernels.h kZBlock)
want_x = np.stack([plan.spectrum(c) for c in clips[:16]])
d = torch.from_numpy(clips).cuda()
d_x = torch.zeros((n_clips, nk, 2), dtype=torch.float32, device="cuda")

ref = handle({"HPFW_FWD_CHUNK": "0"})
ref.stage_spectrum_dev(d.data_ptr(), n, n_clips, d_x.data_ptr())
torch.cuda.synchronize()
x_ref = d_x.cpu().numpy().copy()
assert np.array_equal(x_ref[:16].view(np.uint32), want_x.view(np.uint32)), "the reference run differs from the oracle"
zp, zb = ref.debug_workspace(0)
z_ref = fetch(zp, n_clips * zclip).reshape(n_clips, -1, 2 * hq, 128)
# is the reference itself stable?
for _ in range(2):
    ref.stage_spectrum_dev(d.data_ptr(), n, n_clips, d_x.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_x.cpu().numpy().view(np.uint32), x_ref.view(np.uint32)), "the reference run is not stable"
print(f"# reference ok: n1 {n1} n2 {n2} hq {hq}", file=sys.stderr, flush=True)

CONFIGS = [
    ("default", {}),
    ("v1 lds-staged, chunked 2 streams", {"HPFW_COLS_VARIANT": "1"}),
    ("v1 lds-staged, one launch", {"HPFW_COLS_VARIANT": "1", "HPFW_FWD_CHUNK": "0"}),
    ("v1 lds-staged, chunked 1 stream", {"HPFW_COLS_VARIANT": "1", "HPFW_FWD_STREAMS": "1"}),
    ("default, chunked 1 stream", {"HPFW_FWD_STREAMS": "1"}),
    ("default, chunked 3 streams", {"HPFW_FWD_STREAMS": "3"}),
]
out = []
for name, env in CONFIGS:
    g = handle(env)
    chunked = env.get("HPFW_FWD_CHUNK") != "0"
    lanes = int(env.get("HPFW_FWD_STREAMS", "2"))
    res = {"config": name, "x_bad_clips": [], "z": []}
    for rep in range(reps):
        d_x.zero_()
        g.stage_spectrum_dev(d.data_ptr(), n, n_clips, d_x.data_ptr())
        torch.cuda.synchronize()
        x = d_x.cpu().numpy()
        ne = x.view(np.uint32) != x_ref.view(np.uint32)
        bad = np.nonzero(ne.any(axis=(1, 2)))[0]
        res["x_bad_clips"].append(bad.tolist())
        for c in bad[:3]:
            ks = np.nonzero(ne[c].any(axis=1))[0]
            dv = np.abs(x[c].astype(np.float64) - x_ref[c]).max()
            res.setdefault("x_detail", []).append({"rep": rep, "clip": int(c), "bins": int(ks.size), "rows": np.unique((ks + plan.kmin) % n1).tolist()[:8],
                                                   "max_abs_diff": float(dv), "row_max": float(np.abs(x_ref[c]).max())})
        # z as the call left it: every clip when launched whole, else the last chunk of each stream
        zp, zb = g.debug_workspace(0)
        if not chunked:
            z = fetch(zp, n_clips * zclip).reshape(n_clips, -1, 2 * hq, 128)
            pairs = [(c, z[c]) for c in range(n_clips)]
        else:
            n_chunks = (n_clips + 15) // 16
            pairs = []
            for lane in range(lanes):
                last = max(i for i in range(n_chunks) if i % lanes == lane)
                zl = fetch(zp + lane * 16 * zclip * 4, 16 * zclip).reshape(16, -1, 2 * hq, 128)
                pairs += [(16 * last + j, zl[j]) for j in range(min(16, n_clips - 16 * last))]
        for c, zc in pairs:
            nz = zc.view(np.uint32) != z_ref[c].view(np.uint32)
            if nz.any():
                q1s, planes, cols = np.nonzero(nz)
                res["z"].append({"rep": rep, "clip": int(c), "x_bad": bool(c in bad), "n": int(nz.sum()),
                                 "where": [(int(a), int(b), int(cc), float(zc[a, b, cc]), float(z_ref[c][a, b, cc])) for a, b, cc in
                                           list(zip(q1s, planes, cols))[:6]]})
    g.close()
    print(json.dumps(res), flush=True)
