"""BASELINE.json configs[1] as the bench runs it, and configs[2], [3], [4] at their full index sizes, on one MI355X, against
the oracle.

configs[1]: 30 s clips in the numbers that take the bench's path through the library (at least 96: the forward
            transform in chunks of 16 on two streams, all seven row tiles of the column kernel, the five chirp-z classes
            forked) -- 128 clips on the null stream and on a side stream, every hashprint against the oracle;
configs[2]: 10 000-clip index x 2320 hashprints, queries of 305 in groups of 32 (one group ragged), top-10,
            through all three scan kernels;
configs[3]: the 100 000-clip index as 8 contiguous shards of 12 500 clips -- every shard scanned on this GPU
            with its clip base, the per-shard lists merged by hpfw_gpu_merge_topk (what follows the RCCL
            all-gather) == the unsharded scan == the oracle;
configs[4]: a 125 000-clip shard (1 M / 8), 5 s windows extracted from PCM and searched one at a time
            (hamming_shift_kernel + two-step top-k) and as a batch (grouped matrix-core scan).
MemoryStorage::find (storage.h:27-64) and the notebook's top-10 rule (liveid.ipynb cell 9) define the
expected hits; the oracle is the checker and runs on the host cores."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import hpfw_amd  # noqa: E402
from hpfw_amd import dist as hdist, synth  # noqa: E402

N_HP = 2320
THREADS = min(os.cpu_count() or 8, 64)


def _random_index(seed, n_clips):
    rng = np.random.default_rng(seed)
    db = rng.integers(0, 2 ** 64, size=(n_clips, N_HP), dtype=np.uint64)
    return db, np.arange(n_clips + 1, dtype=np.int64) * N_HP, rng


def _planted(db, rng, specs):
    """specs: (clip, offset, length, bit flips per hashprint); random query when clip < 0"""
    qs = []
    for c, o, k, flips in specs:
        if c < 0:
            qs.append(rng.integers(0, 2 ** 64, size=k, dtype=np.uint64))
            continue
        seg = db[c, o:o + k].copy()
        for _ in range(flips):
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        qs.append(seg)
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    return np.concatenate(qs), q_off


@pytest.fixture(scope="module")
def config2(oracle):
    n_clips = 10000
    db, db_off, rng = _random_index(0xC2, n_clips)
    db[7001] = db[12]                                        # identical clips: ties broken by clip id
    specs = [((q * 131) % n_clips, (q * 37) % (N_HP - 305 + 1), 305, 6) for q in range(64)]
    specs[5] = (12, 100, 305, 0)                             # exact slice of the duplicated clip
    lens = [1, 2, 31, 64, 304, 305, 306, 700, 1500, 2320, 2400] + [int(x) for x in rng.integers(3, 420, 21)]
    for i, k in enumerate(lens):                             # the ragged group
        kk = min(k, N_HP)
        specs.append((-1, 0, k, 0) if i % 4 == 3 or k > N_HP else ((i * 977) % n_clips, (i * 53) % (N_HP - kk + 1), kk, 3))
    q, q_off = _planted(db, rng, specs)
    want = oracle.search_topk(db.ravel(), db_off, q, q_off, 10, n_threads=THREADS)
    return db, db_off, q, q_off, specs, want


def test_config2_full_index_all_scan_paths(gpu, config2, scan_path):
    db, db_off, q, q_off, specs, want = config2
    gpu.index_clear()
    gpu.index_set_clip_base(0)
    gpu.index_add(db.ravel(), db_off)
    got = gpu.search_topk(q, q_off, 10)
    assert np.array_equal(got, want)
    for qi in (0, 17, 63):
        assert got[qi, 0]["clip"] == specs[qi][0] and got[qi, 0]["offset"] == specs[qi][1]
    assert [int(x) for x in got[5]["clip"][:2]] == [12, 7001] and (got[5]["dist"][:2] == 0).all()


def _sharded(gpu, db, db_off, q, q_off, k, n_shards):
    n_clips = db.shape[0]
    per = []
    for r in range(n_shards):
        lo, hi = hdist.shard_range(n_clips, r, n_shards)
        gpu.index_clear()
        gpu.index_set_clip_base(lo)
        gpu.index_add(db[lo:hi].ravel(), db_off[lo:hi + 1] - db_off[lo])
        per.append(gpu.search_topk(q, q_off, k))
    gpu.index_set_clip_base(0)
    return np.stack(per)


def test_config2_eight_shards_equal_unsharded(gpu, config2):
    """the N > 1 data path with real GPU scans: 8 contiguous shards (1250 clips), global clip ids by
    hpfw_gpu_index_set_clip_base, merge of the 8 per-shard top-10 lists == one scan of the whole index"""
    db, db_off, q, q_off, _, want = config2
    per = _sharded(gpu, db, db_off, q, q_off, 10, 8)
    assert np.array_equal(hpfw_amd.merge_topk(per, 10), want)
    # 3 uneven shards, k larger than some shard's useful hits
    assert np.array_equal(hpfw_amd.merge_topk(_sharded(gpu, db, db_off, q, q_off, 10, 3), 10), want)


def test_config3_shards_of_12500(gpu, oracle):
    """configs[3]: 100 000 clips = 8 shards of 12 500; replicated queries; per-shard top-10, merge"""
    n_clips = 100000
    db, db_off, rng = _random_index(0xC3, n_clips)
    db[99999] = db[3]
    specs = [((q * 12347 + 5) % n_clips, (q * 41) % (N_HP - 305 + 1), 305, 8) for q in range(15)] + [(3, 7, 305, 0)]
    q, q_off = _planted(db, rng, specs)
    want = oracle.search_topk(db.ravel(), db_off, q, q_off, 10, n_threads=THREADS)
    gpu.index_clear()
    gpu.index_set_clip_base(0)
    gpu.index_add(db.ravel(), db_off)
    whole = gpu.search_topk(q, q_off, 10)
    assert np.array_equal(whole, want)
    per = _sharded(gpu, db, db_off, q, q_off, 10, 8)
    assert per.shape == (8, 16, 10) and hdist.shard_range(n_clips, 1, 8) == (12500, 25000)
    assert np.array_equal(hpfw_amd.merge_topk(per, 10), want)
    assert [int(x) for x in want[15]["clip"][:2]] == [3, 99999]          # a tie across the first and the last shard
    for qi in range(15):
        assert want[qi, 0]["clip"] == specs[qi][0] and want[qi, 0]["offset"] == specs[qi][1]


def test_config4_shard_streaming_windows(gpu, oracle, filters):
    """configs[4] per GPU: a 125 000-clip shard; 5 s PCM windows -> hashprints -> scan, one at a time (the
    one-query matrix-core kernel and the two-step top-k) and as a batch of 8 (grouped scan)"""
    n_clips = 125000
    db, db_off, rng = _random_index(0xC4, n_clips)
    songs = [synth.gen_clip(700 + i, 30.0) for i in range(4)]
    song_hp = gpu.extract(np.stack(songs))
    where = [5, 40000, 77777, 124999]
    for w, hp in zip(where, song_hp):
        db[w] = hp                                            # four real songs hidden in the random shard
    windows = [synth.gen_query(songs, qi) for qi in range(8)] # noisy 5 s slices: (pcm, song, start sample)
    qhp = gpu.extract(np.stack([w[0] for w in windows]))
    k = qhp.shape[1]
    assert k == 304
    plan = oracle.Plan(windows[0][0].size)
    for i in (0, 5):
        assert np.array_equal(qhp[i], plan.extract(filters, windows[i][0]))
    q_off = np.arange(9, dtype=np.int64) * k
    want = oracle.search_topk(db.ravel(), db_off, qhp.ravel(), q_off, 10, n_threads=THREADS)
    gpu.index_clear()
    gpu.index_set_clip_base(0)
    gpu.index_add(db.ravel(), db_off)
    for i in range(8):                                        # streaming: one window per call
        got = gpu.search_topk(qhp[i], np.array([0, k], np.int64), 10)
        assert np.array_equal(got[0], want[i]), i
    assert np.array_equal(gpu.search_topk(qhp.ravel(), q_off, 10), want)
    hop = 1323000 / 7255 * 3
    for i, (_, song, start) in enumerate(windows):
        assert want[i, 0]["clip"] == where[song] and abs(want[i, 0]["offset"] - start / hop) <= 2


def test_bench_configuration_thirty_second_clips(torch_cuda, oracle, filters):
    """configs[1] as bench.py extracts it: 128 x 30 s clips (n1 = 210; chunks of 16 clips taken in turn by two streams,
    parallel_collector.h:115-137 for every clip) -- on the null stream and on a side stream, twice each so that the second
    call meets the first one's events; every hashprint equals the oracle's"""
    torch = torch_cuda
    n_clips = 128
    base = np.stack([synth.gen_clip(7000 + i, 30.0) for i in range(8)])
    clips = np.concatenate([np.roll(base, 97 * r + 1, axis=1) for r in range(n_clips // 8)])
    n = clips.shape[1]
    plan = oracle.Plan(n)
    assert (plan.n1, plan.n_hp) == (210, N_HP)
    want = plan.extract_batch(filters, clips, n_threads=THREADS)
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    d = torch.from_numpy(clips).cuda()
    side = torch.cuda.Stream()
    for stream in (0, side.cuda_stream, 0, side.cuda_stream):
        hp = torch.zeros((n_clips, plan.n_hp), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        g.extract_dev(d.data_ptr(), n, n_clips, hp.data_ptr(), stream)
        torch.cuda.synchronize()
        got = hp.cpu().numpy().view(np.uint64)
        ne = got != want
        assert not ne.any(), f"stream {stream}: {int(ne.sum())} hashprints differ, clips {np.nonzero(ne.any(axis=1))[0].tolist()[:10]}"
    g.close()
