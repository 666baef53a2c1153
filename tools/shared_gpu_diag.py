"""shared_gpu_diag.py -- where do wrong hashprints come from when two processes share the GPU?
(diagnosis of tests/test_gpu_multi.py::test_two_processes_share_the_gpu; run two of these side by side)

Every repetition runs, on each of several handles that differ in ONE switch, the forward transform alone
(hpfw_gpu_stage_spectrum), the front end alone (hpfw_gpu_stage_spectrogram) and the whole extraction, and compares each
with the oracle: the first stage that differs names the kernel group, the handle that does not differ names the switch.

  python tools/shared_gpu_diag.py <seed> <reps> <seconds> <n_clips> [start_file] [handles: all | default]
(tests/test_gpu_shared.py runs two of these with "default": the product's configuration only)
"""
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

seed, reps, seconds, n_clips = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
start_file = sys.argv[5] if len(sys.argv) > 5 else None
which = sys.argv[6] if len(sys.argv) > 6 else "all"
VARIANTS = [
    ("default", {}),
    ("no_side_streams", {"HPFW_FWD_CHUNK": "0", "HPFW_CQ_SERIAL": "1"}),
    ("cols_lds_staged", {"HPFW_COLS_VARIANT": "1"}),
]
if which == "default":
    VARIANTS = VARIANTS[:1]

filt = synth.make_filters()
n_base = max(1, n_clips // 10)
base = np.stack([synth.gen_clip(seed + i, seconds) for i in range(n_base)])
clips = np.concatenate([np.roll(base, 53 * r, axis=1) for r in range((n_clips + n_base - 1) // n_base)])[:n_clips]
n = clips.shape[1]
plan = oracle.Plan(n)
nk = plan.kmax - plan.kmin
t0 = time.time()
want_hp = plan.extract_batch(filt, clips, n_threads=6)
want_x = np.stack([plan.spectrum(c) for c in clips])
full = seconds <= 10
if full:
    want_db = np.stack([oracle.db(plan.cqmag(x)) for x in want_x])
print(f"# oracle {time.time() - t0:.1f} s; n1 {plan.n1} n2 {plan.n2} kmin {plan.kmin}", file=sys.stderr, flush=True)

handles = []
for name, env in VARIANTS:
    for k, v in env.items():
        os.environ[k] = v
    g = hpfw_amd.Gpu(0)
    for k in env:
        del os.environ[k]
    g.set_filters(filt)
    handles.append((name, g))
d = torch.from_numpy(clips).cuda()
d_hp = torch.zeros((n_clips, plan.n_hp), dtype=torch.int64, device="cuda")
d_x = torch.zeros((n_clips, nk, 2), dtype=torch.float32, device="cuda")
d_db = torch.zeros((n_clips, 121, plan.c), dtype=torch.float32, device="cuda")
# warm every handle (plan upload, workspaces, side streams) before the other process is met
for _, g in handles:
    g.extract_dev(d.data_ptr(), n, n_clips, d_hp.data_ptr())
torch.cuda.synchronize()
if start_file:                       # both processes ready: start together
    open(start_file + f".{seed}", "w").close()
    while len([f for f in os.listdir(os.path.dirname(start_file)) if f.startswith(os.path.basename(start_file))]) < 2:
        time.sleep(0.01)

events = []
totals = {name: {"spectrum": 0, "spectrogram": 0, "hashprints": 0} for name, _ in VARIANTS}
t_start = time.time()
for rep in range(reps):
    for name, g in handles:
        d_x.zero_()
        g.stage_spectrum_dev(d.data_ptr(), n, n_clips, d_x.data_ptr())
        torch.cuda.synchronize()
        x = d_x.cpu().numpy()
        ne = x.view(np.uint32) != want_x.view(np.uint32)
        if ne.any():
            totals[name]["spectrum"] += int(ne.sum())
            for c in np.nonzero(ne.any(axis=(1, 2)))[0]:
                ks = np.nonzero(ne[c].any(axis=1))[0] + plan.kmin
                rows = np.unique(ks % plan.n1)
                events.append({"t": round(time.time() - t_start, 3), "rep": rep, "handle": name, "stage": "spectrum", "clip": int(c),
                               "bins": int(ks.size), "rows_k_mod_n1": rows.tolist()[:40], "n_rows": int(rows.size),
                               "q2_min": int((ks // plan.n1).min()), "q2_max": int((ks // plan.n1).max())})
        if full:
            d_db.zero_()
            g.stage_spectrogram_dev(d.data_ptr(), n, n_clips, d_db.data_ptr())
            torch.cuda.synchronize()
            s = d_db.cpu().numpy()
            ne = s.view(np.uint32) != want_db.view(np.uint32)
            if ne.any():
                totals[name]["spectrogram"] += int(ne.sum())
                for c in np.nonzero(ne.any(axis=(1, 2)))[0]:
                    bands = np.nonzero(ne[c].any(axis=1))[0]
                    cols = np.nonzero(ne[c].any(axis=0))[0]
                    events.append({"t": round(time.time() - t_start, 3), "rep": rep, "handle": name, "stage": "spectrogram", "clip": int(c),
                                   "values": int(ne[c].sum()), "bands": bands.tolist()[:40], "n_bands": int(bands.size),
                                   "col_min": int(cols.min()), "col_max": int(cols.max())})
        d_hp.zero_()
        g.extract_dev(d.data_ptr(), n, n_clips, d_hp.data_ptr())
        torch.cuda.synchronize()
        hp = d_hp.cpu().numpy().view(np.uint64)
        ne = hp != want_hp
        if ne.any():
            totals[name]["hashprints"] += int(ne.sum())
            for c in np.nonzero(ne.any(axis=1))[0]:
                idx = np.nonzero(ne[c])[0]
                events.append({"t": round(time.time() - t_start, 3), "rep": rep, "handle": name, "stage": "hashprints", "clip": int(c),
                               "words": int(idx.size), "first": int(idx.min()), "last": int(idx.max())})
print(json.dumps({"seed": seed, "seconds": seconds, "n_clips": n_clips, "reps": reps, "elapsed_s": round(time.time() - t_start, 2),
                  "totals": totals, "events": events[:400], "n_events": len(events)}))
