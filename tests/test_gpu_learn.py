"""GPU tests of the filter-learning path: frame covariance on f32 MFMA
against a float64 numpy evaluation (tolerance: the reference's own result depends on MKL's
summation order), the host eigen-solve, and extraction with the learned filters against the
oracle (bit-exact: once the filters are fixed the path is the same as everywhere else)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import nsgt_f64  # noqa: E402


def test_covariance_against_float64(torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(21)
    g = hpfw_amd.Gpu(0)
    total = np.zeros((2420, 2420))
    for c, n_clips in ((150, 2), (300, 1), (277, 3)):       # odd frame counts and partial chunks included
        s = rng.uniform(-80, 0, (n_clips, 121, c)).astype(np.float32)
        s[:, ::7, :] *= 0.1
        d_s = torch.from_numpy(s).cuda()
        g.cov_accumulate_db_dev(d_s.data_ptr(), n_clips, c)
        for x in s:
            total += nsgt_f64.covariance(x)
    cov, n_files = g.cov_get()
    assert n_files == 6
    assert np.array_equal(cov, cov.T)
    assert np.abs(cov - total).max() / np.abs(total).max() < 2e-5
    g.cov_reset()
    cov0, n0 = g.cov_get()
    assert n0 == 0 and not cov0.any()
    g.close()


def test_learn_filters_and_extract(torch_cuda, oracle):
    clips = np.stack([synth.gen_clip(700 + i, 3.0) for i in range(6)])
    g = hpfw_amd.Gpu(0)
    with pytest.raises(hpfw_amd.HpfwError):
        g.learn_filters()                                    # nothing accumulated yet
    g.cov_accumulate(clips[:4])
    g.cov_accumulate(clips[4:])
    cov, n_files = g.cov_get()
    assert n_files == 6
    # float64 reference of the whole chain up to the covariance, from the oracle's dB spectrograms
    plan = oracle.Plan(clips.shape[1])
    ref = sum(nsgt_f64.covariance(oracle.db(plan.cqmag(plan.spectrum(c)))) for c in clips)
    assert np.abs(cov - ref).max() / np.abs(ref).max() < 5e-5
    filt = g.learn_filters()
    rows = synth.filters_rows(filt)                          # [64][2420]
    assert np.abs(rows @ rows.T - np.eye(64)).max() < 1e-5
    w, v = np.linalg.eigh(cov.astype(np.float64))
    w, v = w[::-1], v[:, ::-1]
    ray = np.einsum("rk,kl,rl->r", rows, cov.astype(np.float64), rows)
    assert np.abs(ray - w[:64]).max() / w[0] < 1e-5          # Rayleigh quotients = leading eigenvalues
    gaps = (w[:64] - w[1:65]) / w[0]
    sep = gaps > 1e-4                                        # well separated: the vector itself is determined
    dots = np.abs(np.sum(rows * v[:, :64].T, axis=1))
    assert dots[sep].min() > 0.999
    assert (rows[np.arange(64), np.abs(rows).argmax(axis=1)] > 0).all()   # sign convention
    hp = g.extract(clips)
    want = np.stack([plan.extract(filt, c) for c in clips])
    assert np.array_equal(hp, want)
    # the covariance can be saved and restored: same filters
    g2 = hpfw_amd.Gpu(0)
    g2.cov_set(cov, n_files)
    assert np.array_equal(g2.learn_filters(), filt)
    g.close()
    g2.close()


def test_legacy_prepare_learns_and_persists(torch_cuda, oracle, tmp_path):
    clips = [synth.gen_clip(720 + i, 3.0) for i in range(5)]
    paths = []
    for i, c in enumerate(clips):
        p = str(tmp_path / f"song{i}.wav")
        synth.write_wav(p, c)
        paths.append(p)
    cache = str(tmp_path / "cache") + "/"
    pc = hpfw_amd.ParallelCollector()
    pc.load(cache)                                           # nothing there yet: silent (cache.h:77-79)
    res = pc.prepare(paths)                                  # preprocess: covariance -> filters -> save
    assert [n for _, n in res] == [f"song{i}" for i in range(5)]
    raw = open(os.path.join(cache, "filters.cereal"), "rb").read()
    assert raw[:8] == np.array([64, 2420], np.int32).tobytes()
    filt = np.frombuffer(raw[8:], np.float32)
    plan = oracle.Plan(clips[0].size)
    for (hp, _), c in zip(res, clips):
        assert np.array_equal(hp, plan.extract(filt, c))
    rawc = open(os.path.join(cache, "accum_cov.cereal"), "rb").read()
    assert rawc[:8] == np.array([2420, 2420], np.int32).tobytes() and len(rawc) == 8 + 4 * 2420 * 2420
    pc2 = hpfw_amd.ParallelCollector()                       # a new process would do exactly this
    pc2.load(cache)
    assert np.array_equal(pc2.calc_hashprint(paths[2]), res[2][0])


@pytest.mark.parametrize("keep_gb", ["32", "0"])
def test_legacy_prepare_mixed_lengths(torch_cuda, tmp_path, keep_gb, monkeypatch):
    """prepare() over files of several lengths with an unreadable one among them: clips of equal
    length travel together, the results come back in input order and equal the one-file path; with
    no room to keep spectrograms (HPFW_PREPARE_KEEP_GB=0) the second pass reads the files again"""
    monkeypatch.setenv("HPFW_PREPARE_KEEP_GB", keep_gb)
    secs = [3.0, 2.0, 3.0, 4.0, 2.0, 3.0, 2.5]
    paths = []
    for i, sec in enumerate(secs):
        p = str(tmp_path / f"t{i}.wav")
        synth.write_wav(p, synth.gen_clip(900 + i, sec))
        paths.append(p)
    paths.insert(3, str(tmp_path / "nothing_here.wav"))
    open(str(tmp_path / "garbage.wav"), "wb").write(b"RIFF0000WAVEjunk")
    paths.insert(6, str(tmp_path / "garbage.wav"))
    cache = str(tmp_path / "cache") + "/"
    pc = hpfw_amd.ParallelCollector()
    pc.load(cache)
    res = pc.prepare(paths)
    assert [n for _, n in res] == [f"t{i}" for i in range(len(secs))]
    for (hp, name) in res:
        one = pc.calc_hashprint(str(tmp_path / f"{name}.wav"))
        assert np.array_equal(hp, one)
    # the covariance is the sum over the seven readable files, whatever the batching
    rawc = np.frombuffer(open(os.path.join(cache, "accum_cov.cereal"), "rb").read()[8:], np.float32).reshape(2420, 2420)
    g = hpfw_amd.Gpu(0)
    for i in range(len(secs)):
        g.cov_accumulate(synth.gen_clip(900 + i, secs[i])[None, :])
    cov, n_files = g.cov_get()
    g.close()
    assert n_files == len(secs)
    assert np.abs(cov - rawc).max() <= 2e-5 * np.abs(cov).max()


def test_spectrogram_cache_and_incremental_indexing(torch_cuda, oracle, tmp_path):
    """f2: cache/spectros/<stem> (cache.h:30-33, utils.h:77-106) is written by prepare(), and a later prepare()
    -- a new collector, as a new process would hold -- returns hashprints for EVERY cached track under the
    filters it has just learned (collect_fingerprints walks the whole cache, parallel_collector.h:114-137:
    "needed when adding new tracks")."""
    torch = torch_cuda
    secs = [3.0, 4.0, 3.0, 3.0, 5.0]
    clips = [synth.gen_clip(760 + i, s) for i, s in enumerate(secs)]
    paths = []
    for i, c in enumerate(clips):
        p = str(tmp_path / f"trk{i}.wav")
        synth.write_wav(p, c)
        paths.append(p)
    cache = str(tmp_path / "cache") + "/"
    pc = hpfw_amd.ParallelCollector()
    pc.load(cache)
    first = pc.prepare(paths[:3])
    assert [n for _, n in first] == ["trk0", "trk1", "trk2"]
    del pc
    # the files are the reference's cereal image of Matrix<float, 121, Dynamic>: int32 rows, int32 cols, column-major
    g = hpfw_amd.Gpu(0)
    for i in range(3):
        raw = open(os.path.join(cache, "spectros", f"trk{i}"), "rb").read()
        rows, cols = np.frombuffer(raw[:8], np.int32)
        plan = oracle.Plan(clips[i].size)
        assert (rows, cols) == (121, plan.c) and len(raw) == 8 + 4 * 121 * plan.c
        s_file = np.frombuffer(raw[8:], np.float32).reshape(cols, 121)          # [col][bin]: element (b, c) at b + 121 c
        want = oracle.db(plan.cqmag(plan.spectrum(clips[i])))                   # [bin][col]
        assert np.array_equal(s_file.T.view(np.uint32), want.view(np.uint32))
    # a track cached by somebody else (e.g. the reference itself, which allocates one more column when 3 | M)
    rng = np.random.default_rng(9)
    foreign = rng.uniform(-80, 0, (260, 121)).astype(np.float32)
    with open(os.path.join(cache, "spectros", "foreign"), "wb") as f:
        f.write(np.array([121, 260], np.int32).tobytes() + foreign.tobytes())
    open(os.path.join(cache, "spectros", "broken"), "wb").write(b"\\x79\\x00\\x00\\x00junk")   # skipped
    pc2 = hpfw_amd.ParallelCollector()                       # "a new process": loads accum_cov + filters, adds two tracks
    pc2.load(cache)
    second = pc2.prepare(paths[3:])
    names = [n for _, n in second]
    assert names == ["trk3", "trk4", "foreign", "trk0", "trk1", "trk2"]        # this call's files, then the cache, sorted
    filt = np.frombuffer(open(os.path.join(cache, "filters.cereal"), "rb").read()[8:], np.float32)
    g.set_filters(filt)
    by_name = {n: hp for hp, n in second}
    for i, c in enumerate(clips):                            # == extracting all five with those filters
        assert np.array_equal(by_name[f"trk{i}"], oracle.Plan(c.size).extract(filt, c)), i
    assert np.array_equal(by_name["foreign"], g.extract_db(foreign))
    assert np.array_equal(by_name["foreign"], oracle.hashprints_from_db(filt, np.ascontiguousarray(foreign.T)))
    # the covariance kept accumulating across the two runs (parallel_collector.h:93-97 + load())
    rawc = np.frombuffer(open(os.path.join(cache, "accum_cov.cereal"), "rb").read()[8:], np.float32).reshape(2420, 2420)
    for c in clips:
        g.cov_accumulate(c[None, :])
    cov, _ = g.cov_get()
    assert np.abs(cov - rawc).max() <= 2e-5 * np.abs(cov).max()
    g.close()
    # HPFW_NO_SPECTRO_CACHE=1: nothing written, nothing picked up
    os.environ["HPFW_NO_SPECTRO_CACHE"] = "1"
    try:
        third = pc2.prepare(paths[:1])
    finally:
        del os.environ["HPFW_NO_SPECTRO_CACHE"]
    assert [n for _, n in third] == ["trk0"]


def test_covariance_and_filters_of_the_combiner_configuration(torch_cuda, oracle):
    """f3: calc_cov + accumulate + calc_filters (hashprint_handle.h:96-112, parallel_collector.h:93-97) for
    HashprintHandle<uint16_t, MelSpectrogram<>, 32, 50> (frames of 33 x 32 = 1056 values): the covariance summed over
    clips of ragged widths against float64 numpy; the 16 learned filters against the leading eigenvectors; hashprints
    under the learned filters identical to the oracle's"""
    torch = torch_cuda
    cfg = hpfw_amd.COMBINER_CONFIG
    rows, ctx, lag, bits = cfg
    rng = np.random.default_rng(33)
    stride = 420
    cols = np.array([420, 300, 33, 32, 10, 157], np.int32)          # 33: two frames; 32: one frame (adds nothing); 10: none
    base = rng.uniform(-70, -5, (len(cols), rows, 1)).astype(np.float32)
    s = (base + np.cumsum(rng.standard_normal((len(cols), rows, stride)).astype(np.float32), axis=2) * 1.5).astype(np.float32)
    g = hpfw_amd.Gpu(0)
    d_s = torch.from_numpy(s).cuda()
    d_cols = torch.from_numpy(cols).cuda()
    g.cfg_cov_accumulate_dev(cfg, d_s.data_ptr(), d_cols.data_ptr(), 3, stride)
    g.cfg_cov_accumulate_dev(cfg, d_s[3:].data_ptr(), d_cols[3:].data_ptr(), 3, stride)    # accumulates across calls
    cov, n = g.cfg_cov_get(cfg)
    assert n == 6
    want = np.zeros((rows * ctx, rows * ctx))
    for i, c in enumerate(cols):
        nf = c - ctx + 1
        if nf < 2:
            continue
        x = np.stack([s[i, k // ctx, (k % ctx):(k % ctx) + nf] for k in range(rows * ctx)]).astype(np.float64)
        xc = x - x.mean(axis=1, keepdims=True)
        want += xc @ xc.T / (nf - 1)
    assert np.abs(cov - want).max() <= 5e-5 * np.abs(want).max()
    assert np.array_equal(cov, cov.T)
    filt = g.cfg_learn_filters(cfg)                               # column-major [16][1056]
    f_rows = filt.reshape(rows * ctx, bits).T.astype(np.float64)
    w, v = np.linalg.eigh(want)
    w, v = w[::-1], v[:, ::-1]
    ray = np.einsum("rk,kl,rl->r", f_rows, want, f_rows)
    assert np.abs(ray - w[:bits]).max() / w[0] < 1e-4
    assert np.abs(f_rows @ f_rows.T - np.eye(bits)).max() < 1e-4
    nf = stride - ctx + 1
    d_hp = torch.zeros((len(cols), nf - lag), dtype=torch.int16, device="cuda")
    g.cfg_hashprints_dev(cfg, d_s.data_ptr(), d_cols.data_ptr(), len(cols), stride, d_hp.data_ptr(), nf - lag)
    torch.cuda.synchronize()
    hp = d_hp.cpu().numpy().view(np.uint16)
    for i, c in enumerate(cols):
        want_hp = oracle.hashprints_cfg(filt, s[i][:, :c], ctx, lag, bits).astype(np.uint16)
        assert np.array_equal(hp[i][:want_hp.size], want_hp), i
    g.cfg_cov_reset(cfg)
    assert g.cfg_cov_get(cfg)[1] == 0 and not g.cfg_cov_get(cfg)[0].any()
    # the same kernels on the live-id configuration (121 x 20) agree with the lag-correlation kernels of k_cov.hip
    s2 = rng.uniform(-80, 0, (2, 121, 260)).astype(np.float32)
    d_s2 = torch.from_numpy(s2).cuda()
    g.cfg_cov_accumulate_dev((121, 20, 80, 64), d_s2.data_ptr(), 0, 2, 260)
    g.cov_accumulate_db_dev(d_s2.data_ptr(), 2, 260)
    a, _ = g.cfg_cov_get((121, 20, 80, 64))
    b, _ = g.cov_get()
    assert np.abs(a - b).max() <= 5e-5 * np.abs(b).max()
    g.close()
