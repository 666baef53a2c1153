"""interfere.py -- which kernel running BESIDE a stage makes that stage go wrong?  One process, two torch streams:
the victim stage on one, an aggressor on the other, enqueued in turn without host synchronisation; the victim's output
of every round is compared with its output when the GPU was its own.

  [HPFW_GPU_LIB=other build] python tools/interfere.py [rounds]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NC = 48


def handle(env):
    for k, v in env.items():
        os.environ[k] = v
    g = hpfw_amd.Gpu(0)
    for k in env:
        del os.environ[k]
    g.set_filters(filt)
    return g


filt = synth.make_filters()
base = np.stack([synth.gen_clip(4000 + i, 30.0) for i in range(6)])
clips = np.concatenate([np.roll(base, 53 * r, axis=1) for r in range(NC // 6)])
n = clips.shape[1]
plan = oracle.Plan(n)
nk = plan.kmax - plan.kmin
d = torch.from_numpy(clips).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
vic = handle({"HPFW_FWD_CHUNK": "0", "HPFW_CQ_SERIAL": "1"})
agg = handle({"HPFW_FWD_CHUNK": "0", "HPFW_CQ_SERIAL": "1"})
agg_v1 = handle({"HPFW_FWD_CHUNK": "0", "HPFW_CQ_SERIAL": "1", "HPFW_COLS_VARIANT": "1"})

# references, alone on the GPU (the spectrum also against the oracle)
x_ref = torch.zeros((NC, nk, 2), dtype=torch.float32, device="cuda")
vic.stage_spectrum_dev(d.data_ptr(), n, NC, x_ref.data_ptr())
torch.cuda.synchronize()
want = np.stack([plan.spectrum(c) for c in clips[:6]])
assert np.array_equal(x_ref[:6].cpu().numpy().view(np.uint32), want.view(np.uint32))
mag_ref = torch.zeros((NC, 121, plan.c), dtype=torch.float32, device="cuda")
vic.stage_cqmag_dev(x_ref.data_ptr(), n, NC, mag_ref.data_ptr())
db_ref = torch.zeros_like(mag_ref)
vic.stage_db_dev(mag_ref.data_ptr(), NC, plan.c, db_ref.data_ptr())
hp_ref = torch.zeros((NC, plan.n_hp), dtype=torch.int64, device="cuda")
lib = hpfw_amd.lib()
hpfw_amd._lib.check(lib.hpfw_gpu_hashprints_from_db(vic._h, db_ref.data_ptr(), NC, plan.c, hp_ref.data_ptr(), 0))
torch.cuda.synchronize()

big_a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
big_b = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
ma = torch.randn(8192, 8192, dtype=torch.bfloat16, device="cuda")
mb = torch.randn(8192, 8192, dtype=torch.bfloat16, device="cuda")
x_b = torch.zeros_like(x_ref)
mag_b = torch.zeros_like(mag_ref)
hp_b = torch.zeros_like(hp_ref)


def a_none():
    pass


def a_copy():
    with torch.cuda.stream(sb):
        big_b.copy_(big_a)


def a_matmul():
    with torch.cuda.stream(sb):
        torch.mm(ma, mb)


def a_spectrum():
    agg.stage_spectrum_dev(d.data_ptr(), n, NC, x_b.data_ptr(), sb.cuda_stream)


def a_spectrum_v1():
    agg_v1.stage_spectrum_dev(d.data_ptr(), n, NC, x_b.data_ptr(), sb.cuda_stream)


def a_cqmag():
    agg.stage_cqmag_dev(x_ref.data_ptr(), n, NC, mag_b.data_ptr(), sb.cuda_stream)


def a_hashq():
    hpfw_amd._lib.check(lib.hpfw_gpu_hashprints_from_db(agg._h, db_ref.data_ptr(), NC, plan.c, hp_b.data_ptr(), sb.cuda_stream))


# two more LDS-fed matrix kernels of the library as neighbours: the fp4 scan and the f32 projection
rng = np.random.default_rng(5)
idx_hp = rng.integers(0, 2 ** 63, size=(4000, 2320), dtype=np.int64)
agg.index_add(idx_hp.view(np.uint64), np.arange(4001, dtype=np.int64) * 2320)
q_dev = torch.from_numpy(idx_hp[:32, 100:405].copy()).cuda()
q_off = np.arange(33, dtype=np.int64) * 305
hits_b = torch.zeros((32, 10, 4), dtype=torch.int32, device="cuda")
proj_b = torch.zeros((NC, 64, plan.n_frames), dtype=torch.float32, device="cuda")


def a_scan_fp4():
    agg.search_topk_dev(q_dev.data_ptr(), q_off, 10, hits_b.data_ptr(), sb.cuda_stream)


def a_project_f32():
    agg.stage_project_dev(db_ref.data_ptr(), NC, plan.c, proj_b.data_ptr(), sb.cuda_stream)


outs_x = [torch.zeros_like(x_ref) for _ in range(3)]
outs_m = [torch.zeros_like(mag_ref) for _ in range(3)]
outs_h = [torch.zeros_like(hp_ref) for _ in range(3)]


def v_spectrum(i):
    vic.stage_spectrum_dev(d.data_ptr(), n, NC, outs_x[i % 3].data_ptr(), sa.cuda_stream)
    return outs_x[i % 3], x_ref


def v_cqmag(i):
    vic.stage_cqmag_dev(x_ref.data_ptr(), n, NC, outs_m[i % 3].data_ptr(), sa.cuda_stream)
    return outs_m[i % 3], mag_ref


def v_hashq(i):
    hpfw_amd._lib.check(lib.hpfw_gpu_hashprints_from_db(vic._h, db_ref.data_ptr(), NC, plan.c, outs_h[i % 3].data_ptr(), sa.cuda_stream))
    return outs_h[i % 3], hp_ref


AGG = [("none", a_none), ("hbm copy 1 GB", a_copy), ("bf16 matmul 8192^3", a_matmul), ("spectrum (q3 cols + rows)", a_spectrum),
       ("spectrum (lds-staged cols + rows)", a_spectrum_v1), ("cqmag", a_cqmag), ("hashprint_q", a_hashq),
       ("fp4 scan (hamming_mfma_kernel)", a_scan_fp4), ("f32 projection (project_kernel)", a_project_f32)]
VIC = [("spectrum", v_spectrum), ("cqmag", v_cqmag), ("hashprint_q", v_hashq)]
print("# lib", hpfw_amd._lib.LIB_PATH, file=sys.stderr)
for vn, vf in VIC:
    for an, af in AGG:
        bad_rounds, bad_clips = 0, 0
        for base_i in range(0, rounds, 3):
            pend = []
            for i in range(base_i, min(rounds, base_i + 3)):
                af()
                pend.append(vf(i))
                af()
            torch.cuda.synchronize()
            for got, ref in pend:
                ne = (got.view(torch.int32 if got.dtype == torch.float32 else got.dtype) != ref.view(torch.int32 if ref.dtype == torch.float32 else ref.dtype)).reshape(NC, -1).any(dim=1)
                k = int(ne.sum())
                bad_rounds += k > 0
                bad_clips += k
        print(json.dumps({"victim": vn, "aggressor": an, "rounds": rounds, "bad_rounds": bad_rounds, "bad_clips": bad_clips}), flush=True)
