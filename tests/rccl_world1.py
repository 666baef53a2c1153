"""Run by test_gpu_dist.py in a child process: the collectives of the N > 1 path on the "nccl" (RCCL)
backend with device tensors, world size 1 -- all a one-GPU box can hold (RCCL refuses two ranks on one
device); the world-size-2 logic is covered on gloo by test_dist_cpu.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpfw_amd  # noqa: E402
from hpfw_amd import dist as hdist, synth  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # bench.py's max-over-ranks timing
assert float(t.item()) == 3.5

hits = np.zeros((5, 3), hpfw_amd.HIT_DTYPE)                    # per-shard top-k -> all-gather -> merge
hits["dist"] = np.arange(15).reshape(5, 3) * 7 % 11
hits["clip"] = np.arange(15).reshape(5, 3)
hits["offset"] = 100 + np.arange(15).reshape(5, 3)
order = np.lexsort((hits["clip"], hits["dist"]), axis=1)
want = np.take_along_axis(hits, order, axis=1)
got = hdist.allgather_topk(want, 3, device=dev)
assert np.array_equal(got, want), (got, want)

gathered = torch.empty((1, 5, 3, 4), dtype=torch.int32, device=dev)   # the call bench.py makes
local = torch.from_numpy(want.view(np.int32).reshape(5, 3, 4).copy()).to(dev)
dist.all_gather_into_tensor(gathered, local)
assert torch.equal(gathered[0], local)

g = hpfw_amd.Gpu(0)                                            # sharded filter learning
g.cov_accumulate(np.stack([synth.gen_clip(40 + i, 3.0) for i in range(3)]))
cov, n = g.cov_get()
filt = hdist.learn_filters_sharded(g, device=dev)
g2 = hpfw_amd.Gpu(0)
g2.cov_set(np.triu(cov) + np.triu(cov, 1).T, n)
assert np.array_equal(filt, g2.learn_filters())
g.close()
g2.close()
dist.destroy_process_group()
print("rccl world-1 ok")
