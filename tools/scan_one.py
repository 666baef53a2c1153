"""One-query scans over a 125 000-clip random index (configs[4]'s shard), for rocprofv3.  python tools/scan_one.py [reps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import hpfw_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_clips, per, k = 125000, 2320, 305
g = hpfw_amd.Gpu(0)
db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_clips * per,), dtype=torch.int64, device="cuda")
off = np.arange(0, (n_clips + 1) * per, per, dtype=np.int64)
g.index_add_dev(db.data_ptr(), off, 0)
q = db[777 * per + 100: 777 * per + 100 + k].cpu().numpy().view(np.uint64)
q_off = np.array([0, k], np.int64)
hits = g.search_topk(q, q_off, 10)
assert hits[0, 0]["clip"] == 777 and hits[0, 0]["offset"] == 100 and hits[0, 0]["dist"] == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.search_topk(q, q_off, 10)
dt = (time.perf_counter() - t0) / reps
print(f"one query of {k} hashprints against {n_clips} clips: {dt * 1e3:.3f} ms per search")
