"""One query against a resident shard: time of the scan kernels and a check of the hit (run on the GPU box).
   python tools/scan_one.py [clips] [query hashprints] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hpfw_amd

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
kq = int(sys.argv[2]) if len(sys.argv) > 2 else 304
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n_hp = 2320
gpu = hpfw_amd.Gpu(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_clips, n_hp), dtype=torch.int64, generator=g, device=dev)
stream = torch.cuda.current_stream().cuda_stream
gpu.index_clear()
gpu.index_add_dev(db.data_ptr(), np.arange(0, (n_clips + 1) * n_hp, n_hp, dtype=np.int64), stream)
torch.cuda.synchronize()
hits = torch.empty((1, 5, 4), dtype=torch.int32, device=dev)
q_off = np.array([0, kq], dtype=np.int64)
wrong = 0
scan_idx = hpfw_amd.KERNEL_KINDS.index("hamming_scan")
gpu.set_kernel_timing(1 << scan_idx)
lat = []
for r in range(rounds):
    clip, off = (r * 7919 + 13) % n_clips, (r * 131) % (n_hp - kq + 1)
    q = db[clip, off:off + kq].clone()
    q[::7] ^= 1 << (r % 62)                       # a few flipped bits
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu.search_topk_dev(q.data_ptr(), q_off, 5, hits.data_ptr(), stream)
    res = hits.cpu().numpy().reshape(-1).view(hpfw_amd.HIT_DTYPE).reshape(1, 5)
    lat.append((time.perf_counter() - t0) * 1e3)
    top = res[0, 0]
    wrong += int(top["clip"] != clip or top["offset"] != off)
kt = gpu.kernel_timing()
print({"clips": n_clips, "kq": kq, "rounds": rounds, "wrong": wrong, "latency_ms_p50": round(float(np.median(lat)), 3),
       "scan": kt.get("hamming_scan"), "pairs_per_s": None if not kt.get("hamming_scan") else
       round(n_clips * (n_hp - kq + 1) * kq / (kt["hamming_scan"][0] / max(kt["hamming_scan"][1], 1) * 1e-3), 1)})
