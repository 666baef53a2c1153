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


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_search_equals_single_rank(world):
    """(3 ranks: shards of unequal size; 4: the next step of the driver's N = 1, 2, 4, 8 ladder)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import hpfw_amd
    want = np.load(os.path.join(ROOT, "tests", "golden", "search.npz"))["top5"]
    from hpfw_amd import dist as hdist
    ranges = sorted(r[2] for r in got)
    assert ranges == [hdist.shard_range(8, r, world) for r in range(world)] and ranges[0][0] == 0 and ranges[-1][1] == 8
    for _, raw, _ in got:
        merged = np.frombuffer(raw, hpfw_amd.HIT_DTYPE).reshape(want.shape)
        assert np.array_equal(merged, want)


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench_search's shape in small: every rank holds n_local random clips, the queries are cut from shard 0's clips
    n_local, n_hp, kq, nq, topk = 24, 200, 30, 40, 5
    def shard(r):
        return np.random.default_rng(0x1D8 + r).integers(0, 2 ** 64, (n_local, n_hp), dtype=np.uint64)
    db, db0 = shard(rank), shard(0)
    src = np.arange(nq) % n_local
    offs = (np.arange(nq) * 37) % (n_hp - kq + 1)
    qq = np.stack([db0[src[i], offs[i]:offs[i] + kq] for i in range(nq)])
    qq ^= np.uint64(1) << np.random.default_rng(0x51).integers(0, 63, qq.shape, dtype=np.uint64)   # a bit flip per hashprint
    db_off = np.arange(0, (n_local + 1) * n_hp, n_hp, dtype=np.int64)
    q_off = np.arange(0, (nq + 1) * kq, kq, dtype=np.int64)
    local = oracle.search_topk(db.ravel(), db_off, qq.ravel(), q_off, topk)          # stands in for the GPU scan of the shard
    local["clip"][local["clip"] != 0xFFFFFFFF] += rank * n_local                     # hpfw_gpu_index_set_clip_base
    hits = torch.from_numpy(local.view(np.int32).reshape(nq, topk, 4).copy())
    gathered = torch.empty((world, nq, topk, 4), dtype=torch.int32)
    bench.exchange_hits(dist, hits, gathered, world, cpu_collectives=True)           # bench.py's own exchange step
    res = bench.merged_hits(hits, gathered, world, nq, topk)                         # ... and merge
    whole = np.concatenate([shard(r) for r in range(world)])
    want = oracle.search_topk(whole.ravel(), np.arange(0, (world * n_local + 1) * n_hp, n_hp, dtype=np.int64), qq.ravel(), q_off, topk)
    q.put((rank, bool(np.array_equal(res, want)), bench.planted_found(res, nq, n_local, n_hp, kq)))
    dist.destroy_process_group()


def test_bench_search_exchange_and_merge_two_ranks():
    """bench.py's N > 1 search leg with gloo in place of RCCL: its exchange step and merge, fed the per-shard lists of two
    ranks, give the unsharded answer on both ranks and find every planted query"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, True), (1, True, True)]


class _HostLearner:
    """Stands in for hpfw_amd.Gpu in the build container: holds a covariance on the host and solves
    with the product's own host eigen-solver (hpfw_gpu_host_top_eigenvectors needs no GPU)."""

    def __init__(self, cov, n_files):
        self.cov, self.n, self.filters = cov, n_files, None

    def cov_get(self):
        return self.cov.copy(), self.n

    def cov_set(self, cov, n_files):
        self.cov, self.n = np.array(cov, np.float32), n_files

    def learn_filters(self):
        import ctypes
        import hpfw_amd
        nn = self.cov.shape[0]
        rows = np.zeros((64, nn), np.float32)
        rc = hpfw_amd.lib().hpfw_gpu_host_top_eigenvectors(self.cov.ctypes.data_as(ctypes.c_void_p), nn, 64,
                                                           rows.ctypes.data_as(ctypes.c_void_p), None)
        assert rc == 0
        return np.ascontiguousarray(rows.T).ravel()          # column-major [64][nn]

    def set_filters(self, f):
        self.filters = np.array(f)


def _shard_cov(rank):
    rng = np.random.default_rng(100 + rank)
    x = rng.standard_normal((2420, 400)).astype(np.float32) * np.linspace(3, 0.1, 2420, dtype=np.float32)[:, None]
    return (x @ x.T / 399).astype(np.float32)


def _learn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hpfw_amd import dist as hdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _HostLearner(_shard_cov(rank), 3 + rank)
    f = hdist.learn_filters_sharded(g)
    q.put((rank, f.tobytes(), g.filters.tobytes(), g.n, g.cov.tobytes()))
    dist.destroy_process_group()


def test_sharded_filter_learning():
    """covariances of the shards are summed, rank 0 solves, every rank ends with the same filters"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_learn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == got[0][2] == got[1][2]
    assert got[0][3] == got[1][3] == 7
    total = np.frombuffer(got[0][4], np.float32).reshape(2420, 2420)
    want = _shard_cov(0) + _shard_cov(1)
    assert np.abs(total - want).max() <= 1e-6 * np.abs(want).max()
    rows = np.frombuffer(got[0][1], np.float32).reshape(2420, 64).T
    w = np.linalg.eigvalsh(want.astype(np.float64))[::-1]
    ray = np.einsum("rk,kl,rl->r", rows, want.astype(np.float64), rows)
    assert np.abs(ray - w[:64]).max() / w[0] < 1e-5


def test_shard_range_covers_everything():
    from hpfw_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100000, 12345):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
