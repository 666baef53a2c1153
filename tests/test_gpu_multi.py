"""The native multi-GPU host path (include/hpfw_gpu_multi.h, libhpfw_gpu_multi.so: C++ over the C-ABI + RCCL)
on the one GPU this box has: a group of one shard (ncclCommInitAll of world size 1, a real ncclAllGather /
ncclAllReduce) and groups of several shards placed on device 0 (their lists travel in one all-gather of the
device's send buffer).  Sharded results must equal the unsharded ones and the oracle's
(identical at any number of shards)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import hpfw_amd  # noqa: E402
from hpfw_amd import _lib, multi, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    """visible GPUs, counted without initialising the runtime (torch.cuda.device_count() does not, on this image)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def _device_sets():
    """shard placements: one GPU carries every shard (what the one-GPU box can run); with more than one GPU visible
    the real cross-device path -- ncclCommInitAll over several devices, a grouped ncclAllGather across their streams,
    an in-place ncclAllReduce on each device's own covariance -- on two devices and on all of them.  On a one-GPU box
    those cases are listed as skipped, not silently absent."""
    sets = [[0], [0, 0], [0, 0, 0, 0, 0, 0, 0, 0]]
    n = _n_devices()
    multi_sets = [[0, 1]] + ([list(range(n))] if n > 2 else [])
    for d in multi_sets:
        marks = [] if n >= len(set(d)) else [pytest.mark.skip(reason=f"needs {len(set(d))} GPUs, {n} visible")]
        sets.append(pytest.param(d, marks=marks, id="devices-" + "-".join(map(str, d))))
    if n > 2:                                                   # two shards per device over all devices
        sets.append(pytest.param(list(range(n)) * 2, id=f"two-shards-on-each-of-{n}"))
    return sets


def _ragged(rng, lens):
    hp = [rng.integers(0, 2 ** 64, size=n, dtype=np.uint64) for n in lens]
    return np.concatenate(hp), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


@pytest.mark.parametrize("devices", _device_sets())
def test_group_search_equals_unsharded_and_oracle(torch_cuda, gpu, oracle, devices):
    rng = np.random.default_rng(31)
    lens = [int(x) for x in rng.integers(1, 900, 203)] + [2320, 0, 5]
    db, db_off = _ragged(rng, lens)
    db[db_off[200]:db_off[200] + min(lens[200], lens[3])] = db[db_off[3]:db_off[3] + min(lens[200], lens[3])]
    qs = []
    for i in range(45):
        c = (i * 17) % len(lens)
        k = min(lens[c], int(rng.integers(1, 400)))
        if k and i % 4:
            o = int(rng.integers(0, lens[c] - k + 1))
            seg = db[db_off[c] + o: db_off[c] + o + k].copy()
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        else:
            seg = rng.integers(0, 2 ** 64, size=max(k, 1), dtype=np.uint64)
        qs.append(seg)
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    q = np.concatenate(qs)
    want = oracle.search_topk(db, db_off, q, q_off, 10, n_threads=8)
    gpu.index_clear()
    gpu.index_set_clip_base(0)
    gpu.index_add(db, db_off)
    assert np.array_equal(gpu.search_topk(q, q_off, 10), want)
    g = multi.GpuGroup(devices)
    assert g.shards == len(devices)
    assert g.exchange == ("rccl" if len(set(devices)) == len(devices) else "rccl+local")
    g.index_build(db, db_off)
    for k in (1, 10):
        assert np.array_equal(g.search_topk(q, q_off, k), want[:, :k])
    # one query, and a rebuild with fewer clips than shards (some shards empty)
    assert np.array_equal(g.search_topk(qs[1], np.array([0, qs[1].size], np.int64), 10), want[1:2])
    g.index_build(db[:db_off[3]], db_off[:4])
    small = oracle.search_topk(db[:db_off[3]], db_off[:4], q, q_off, 4, n_threads=8)
    assert np.array_equal(g.search_topk(q, q_off, 4), small)
    g.close()


def test_group_extraction_and_learning(torch_cuda, oracle, filters):
    clips = np.stack([synth.gen_clip(810 + i, 3.0) for i in range(7)])
    plan = oracle.Plan(clips.shape[1])
    want = plan.extract_batch(filters, clips, n_threads=7)
    one = hpfw_amd.Gpu(0)
    one.cov_accumulate(clips)
    f_one = one.learn_filters()
    cov_one, n_one = one.cov_get()
    n_dev = _n_devices()
    placements = [[0], [0, 0, 0]] + ([[0, 1], list(range(n_dev))] if n_dev > 1 else [])
    for devices in placements:
        g = multi.GpuGroup(devices)
        g.set_filters(filters)
        assert np.array_equal(g.extract(clips, plan.n_hp), want)          # clips sharded, order kept
        g.cov_reset()
        g.cov_accumulate(clips)                                           # every shard: the covariance of its block
        f = g.learn_filters()                                             # all-reduce (or host sum), solve, install
        if len(devices) == 1:
            assert np.array_equal(f, f_one)                               # a sum of one: the same bits
        cov_g, n_g = hpfw_amd.Gpu.from_handle(g.handle(0)).cov_get()      # the summed covariance, on shard 0
        assert n_g == n_one and np.abs(cov_g - cov_one).max() <= 2e-5 * np.abs(cov_one).max()
        rows = f.reshape(2420, 64).T.astype(np.float64)
        w = np.linalg.eigvalsh(cov_one.astype(np.float64))[::-1]
        ray = np.einsum("rk,kl,rl->r", rows, cov_one.astype(np.float64), rows)
        assert np.abs(ray - w[:64]).max() / w[0] < 1e-4
        assert np.abs(rows @ rows.T - np.eye(64)).max() < 1e-4
        got = g.extract(clips, plan.n_hp)                                 # the learned filters are installed everywhere
        assert np.array_equal(got, plan.extract_batch(f, clips, n_threads=7))
        g.close()
    one.close()


def test_group_learning_keeps_accumulating(torch_cuda):
    """accumulate A, learn, accumulate B, learn (the reference's accum_cov grows across prepare() calls,
    parallel_collector.h:93-97): the second result is the single handle's on A + B, i.e. the first sum entered the
    second one once and not once per shard"""
    a = np.stack([synth.gen_clip(870 + i, 3.0) for i in range(5)])
    b = np.stack([synth.gen_clip(880 + i, 3.0) for i in range(4)])
    one = hpfw_amd.Gpu(0)
    one.cov_accumulate(a)
    one.learn_filters()
    one.cov_accumulate(b)
    f_one = one.learn_filters()
    cov_one, n_one = one.cov_get()
    assert n_one == 9
    n_dev = _n_devices()
    for devices in [[0], [0, 0, 0]] + ([[0, 1], list(range(n_dev))] if n_dev > 1 else []):
        g = multi.GpuGroup(devices)
        g.cov_reset()
        g.cov_accumulate(a)
        g.learn_filters()
        g.cov_accumulate(b)
        f = g.learn_filters()
        cov_g, n_g = hpfw_amd.Gpu.from_handle(g.handle(0)).cov_get()
        assert n_g == 9
        assert np.abs(cov_g - cov_one).max() <= 2e-5 * np.abs(cov_one).max(), devices
        if len(devices) == 1:
            assert np.array_equal(f, f_one)
        rows = f.reshape(2420, 64).T.astype(np.float64)
        w = np.linalg.eigvalsh(cov_one.astype(np.float64))[::-1]
        ray = np.einsum("rk,kl,rl->r", rows, cov_one.astype(np.float64), rows)
        assert np.abs(ray - w[:64]).max() / w[0] < 1e-4, devices
        g.close()
    one.close()


def test_cpp_sharded_live_song_identification(torch_cuda, filters, tmp_path):
    """LiveSongIdentification<GpuCollector, ShardedGpuStorage>: the same stdout as the single-GPU storage"""
    libdir = os.path.dirname(_lib.LIB_PATH)
    exes = {}
    for name, libs in (("live_id", ["-lhpfw_gpu"]), ("live_id_multi", ["-lhpfw_gpu_multi", "-lhpfw_gpu"])):
        exes[name] = str(tmp_path / name)
        r = subprocess.run(["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", name + ".cpp"), "-o", exes[name], "-L", libdir] + libs +
                           ["-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    clips = [synth.gen_clip(830 + i, 8.0) for i in range(5)]
    tracks, queries = [], []
    for i, c in enumerate(clips):
        p = str(tmp_path / f"tune{i}.wav")
        synth.write_wav(p, c)
        tracks.append(p)
    for qi in range(4):
        pcm, ci, start = synth.gen_query(clips, qi, seconds=3.0)
        p = str(tmp_path / f"live_tune{ci}_{qi}.wav")
        synth.write_wav(p, pcm)
        queries.append(p)
    outs = {}
    for name, env, extra in (("live_id", {}, []), ("live_id_multi", {"HPFW_GPU_DEVICES": "0,0,0"}, []),
                             ("live_id_multi", {"HPFW_GPU_DEVICES": "0,0"}, ["--one-collector"])):
        work = tmp_path / ("run_" + name + "_".join(extra))
        os.makedirs(str(work / "cache"))
        with open(str(work / "cache" / "filters.cereal"), "wb") as f:
            f.write(np.array([64, 2420], np.int32).tobytes() + np.ascontiguousarray(filters, np.float32).tobytes())
        r = subprocess.run([exes[name]] + extra + ["--index"] + tracks + ["--search"] + queries, cwd=str(work), capture_output=True,
                           text=True, timeout=300, env=dict(os.environ, HPFW_PREPARE_KEEP_FILTERS="1", **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name + "".join(extra)] = [ln for ln in r.stdout.splitlines() if ln.startswith("=> ")]
        if name == "live_id_multi":
            assert ("shards: 2" if extra else "shards: 3") in r.stderr
    assert outs["live_id"] == outs["live_id_multi"] == outs["live_id_multi--one-collector"] and outs["live_id"][-1] == "=> 0 1"


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]] + [d for d in _device_sets()[3:]])
def test_group_prepare_learns_over_shards(torch_cuda, oracle, tmp_path, devices):
    """ParallelCollector::prepare over the shards (hpfw_gpu_group_prepare): files sharded per device, covariance summed
    (ncclAllReduce, or on the host when shards share a device), filters solved once, every shard hashes its own files;
    a second call adds tracks and returns every track of the cache"""
    secs = [3.0, 4.0, 3.0, 3.5, 3.0, 5.0, 3.0]
    clips = [synth.gen_clip(850 + i, s) for i, s in enumerate(secs)]
    paths = []
    for i, c in enumerate(clips):
        p = str(tmp_path / f"song{i}.wav")
        synth.write_wav(p, c)
        paths.append(p)
    bad = str(tmp_path / "missing.wav")
    cache = str(tmp_path / "cache") + "/"
    g = multi.GpuGroup(devices)
    g.load(cache)                                             # nothing there yet
    res = g.prepare(paths[:3] + [bad] + paths[3:5])
    assert [n for _, n in res] == [f"song{i}" for i in range(5)]
    filt = np.frombuffer(open(os.path.join(cache, "filters.cereal"), "rb").read()[8:], np.float32)
    for (hp, _), c in zip(res, clips[:5]):
        assert np.array_equal(hp, oracle.Plan(c.size).extract(filt, c))
    assert sorted(os.listdir(os.path.join(cache, "spectros"))) == [f"song{i}" for i in range(5)]
    # the covariance in the cache is the sum over the five readable files, whatever the sharding
    one = hpfw_amd.Gpu(0)
    for c in clips[:5]:
        one.cov_accumulate(c[None, :])
    cov, _ = one.cov_get()
    rawc = np.frombuffer(open(os.path.join(cache, "accum_cov.cereal"), "rb").read()[8:], np.float32).reshape(2420, 2420)
    assert np.abs(cov - rawc).max() <= 2e-5 * np.abs(cov).max()
    devices = list(devices)
    if len(devices) == 1:                                     # one shard: the single collector's bits
        pc = hpfw_amd.ParallelCollector()
        cache1 = str(tmp_path / "cache1") + "/"
        pc.load(cache1)
        ref = pc.prepare(paths[:5])
        assert all(np.array_equal(a[0], b[0]) for a, b in zip(res, ref))
    assert np.array_equal(g.calc_hashprint(paths[1]), res[1][0])
    g.close()
    g2 = multi.GpuGroup(devices)                              # "a new process" adds two tracks
    g2.load(cache)
    res2 = g2.prepare(paths[5:])
    assert [n for _, n in res2] == ["song5", "song6"] + [f"song{i}" for i in range(5)]
    filt2 = np.frombuffer(open(os.path.join(cache, "filters.cereal"), "rb").read()[8:], np.float32)
    by_name = {n: hp for hp, n in res2}
    for i, c in enumerate(clips):
        assert np.array_equal(by_name[f"song{i}"], oracle.Plan(c.size).extract(filt2, c)), i
    for c in clips[5:]:
        one.cov_accumulate(c[None, :])
    cov2, _ = one.cov_get()
    rawc2 = np.frombuffer(open(os.path.join(cache, "accum_cov.cereal"), "rb").read()[8:], np.float32).reshape(2420, 2420)
    assert np.abs(cov2 - rawc2).max() <= 2e-5 * np.abs(cov2).max()
    one.close()
    g2.close()
