"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The path shards per audio file: a rank extracts and indexes a
contiguous block of clips and searches only its shard; the single exchange step is an all-gather
of the per-shard top-k lists followed by the same deterministic merge on every rank
(SURVEY.md section 8(e)).  Extraction needs no collective at all."""
import os

import numpy as np

from . import _lib


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when run directly"""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(n_items, rank, world):
    """contiguous block [lo, hi) of rank: sizes differ by at most one, earlier ranks get the extra"""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_topk(local_hits, k, group=None, device=None):
    """local_hits: [n_q][k] HIT_DTYPE with GLOBAL clip ids.  Returns the merged [n_q][k] list,
    identical on every rank.  16 bytes per hit: 160 KB per rank for 1000 queries, k = 10."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    a = np.ascontiguousarray(local_hits, _lib.HIT_DTYPE)
    n_q = a.shape[0]
    t = torch.from_numpy(a.view(np.int32).reshape(n_q, k, 4).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group) if hasattr(dist, "all_gather_into_tensor") and t.is_cuda \
        else _all_gather_list(out, t, group)
    per_shard = out.cpu().numpy().reshape(world, n_q, k * 4).view(_lib.HIT_DTYPE).reshape(world, n_q, k)
    return _lib.merge_topk(per_shard, k)


def _all_gather_list(out, t, group):
    import torch.distributed as dist
    parts = [out[i] for i in range(out.shape[0])]
    dist.all_gather(parts, t, group=group)


def learn_filters_sharded(gpu, group=None, device=None):
    """Filter learning over ranks (reference parallel_collector.h:82-112 accumulates one accum_cov
    over all files): every rank has accumulated the covariance of its own shard of clips on `gpu`;
    the matrices and the file counts are summed with one all-reduce (23 MB), rank 0 solves for the
    64 leading eigenvectors and broadcasts them, and every rank installs the same filters.
    Returns the filters (flat, column-major [64][2420])."""
    import torch
    import torch.distributed as dist
    cov, n_files = gpu.cov_get()
    t = torch.from_numpy(cov)
    cnt = torch.tensor([n_files], dtype=torch.int64)
    if device is not None:
        t, cnt = t.to(device), cnt.to(device)
    dist.all_reduce(t, group=group)
    dist.all_reduce(cnt, group=group)
    total = t.cpu().numpy()
    total = np.triu(total) + np.triu(total, 1).T          # exactly symmetric whatever the reduction order
    gpu.cov_set(total, int(cnt.item()))
    f = torch.zeros(64 * 2420, dtype=torch.float32)
    if dist.get_rank(group) == 0:
        f = torch.from_numpy(gpu.learn_filters())
    if device is not None:
        f = f.to(device)
    dist.broadcast(f, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    filt = f.cpu().numpy()
    gpu.set_filters(filt)
    return filt
