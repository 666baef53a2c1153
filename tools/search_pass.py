"""One search over a synthetic index resident in HBM -- the profiling workload of the scan
(rocprofv3 ... -- python3 tools/search_pass.py [clips] [queries] [passes])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = hpfw_amd.Gpu(0)
db = synth.random_hashprints(n_clips, 2320)
g.index_add(db.ravel(), np.arange(n_clips + 1, dtype=np.int64) * 2320)
q, ids, offs = synth.planted_queries(db, n_q, 305)
d_q = torch.from_numpy(q.view(np.int64)).cuda()
q_off = np.arange(n_q + 1, dtype=np.int64) * 305
d_out = torch.zeros(n_q * 10 * 4, dtype=torch.int32, device="cuda")
for _ in range(passes):
    t0 = time.perf_counter()
    g.search_topk_dev(d_q.data_ptr(), q_off, 10, d_out.data_ptr())
    torch.cuda.synchronize()
    print(f"search: {(time.perf_counter() - t0) * 1e3:.1f} ms")
hits = d_out.cpu().numpy().view(hpfw_amd.HIT_DTYPE).reshape(n_q, 10)
print("planted found:", bool((hits["clip"][:, 0] == ids).all() and (hits["offset"][:, 0] == offs).all()))
