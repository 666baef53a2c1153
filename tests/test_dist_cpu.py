"""The N > 1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous shard of the clips,
computes its local top-k with GLOBAL clip ids (the oracle stands in for the GPU scan here), the
ranks all-gather the per-shard lists and run the product's deterministic merge
(hpfw_amd.dist.allgather_topk -> hpfw_gpu_merge_topk).  The result must equal the single-rank
answer on every rank (identical at any number of GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hpfw_amd import dist as hdist
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "search.npz"))
    db, db_off, qq, q_off = g["db"], g["db_off"], g["q"], g["q_off"]
    n_clips = len(db_off) - 1
    lo, hi = hdist.shard_range(n_clips, rank, world)
    local_off = db_off[lo:hi + 1] - db_off[lo]
    local = oracle.search_topk(db[db_off[lo]:db_off[hi]], local_off, qq, q_off, 5)
    valid = local["clip"] != 0xFFFFFFFF
    local["clip"][valid] += lo                               # what hpfw_gpu_index_set_clip_base does
    merged = hdist.allgather_topk(local, 5)
    q.put((rank, merged.tobytes(), (lo, hi)))
    dist.destroy_process_group()


def test_sharded_search_equals_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import hpfw_amd
    want = np.load(os.path.join(ROOT, "tests", "golden", "search.npz"))["top5"]
    ranges = sorted(r[2] for r in got)
    assert ranges == [(0, 4), (4, 8)]
    for _, raw, _ in got:
        merged = np.frombuffer(raw, hpfw_amd.HIT_DTYPE).reshape(want.shape)
        assert np.array_equal(merged, want)


def test_shard_range_covers_everything():
    from hpfw_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100000, 12345):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
