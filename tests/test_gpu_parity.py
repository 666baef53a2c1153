"""GPU parity: every stage of the HIP path against the CPU oracle on the same seeded inputs,
called through the C-ABI (include/hpfw_gpu.h).  Bars: bit-exact for hashprints, Hamming distances,
offsets and top-k order; the float stages are also required to be bit-exact against the oracle
(both implement DESIGN.md's arithmetic specification) and within 1e-4 of the maximum against the
float64 definition (oracle/nsgt_f64.py) -- the tolerance BASELINE.json's north_star states for the
CQT magnitudes."""
import numpy as np
import pytest

from conftest import bits_equal, ulp_diff

pytestmark = pytest.mark.gpu

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _report(name, got, want):
    d = ulp_diff(got, want)
    return (f"{name}: {int((d > 0).sum())} of {d.size} values differ, max {int(d.max())} ulp, "
            f"max abs {float(np.abs(got.astype(np.float64) - want).max()):.3e}")


@pytest.mark.parametrize("seconds", [3.0, 5.0, 30.0])
def test_stages_bit_exact(gpu, torch_cuda, oracle, filters, seconds):
    torch = torch_cuda
    n_clips = 3 if seconds < 30 else 2
    clips = np.stack([synth.gen_clip(100 + i, seconds) for i in range(n_clips)])
    n = clips.shape[1]
    plan = oracle.Plan(n)
    g = gpu.geometry(n)
    assert (g.n1, g.n2, g.kmin, g.kmax, g.m, g.c, g.n_frames, g.n_hp) == (
        plan.n1, plan.n2, plan.kmin, plan.kmax, plan.m, plan.c, plan.n_frames, plan.n_hp)
    nk = plan.kmax - plan.kmin
    d_pcm = _dev(torch, clips)
    # a1 + forward DFT
    d_x = torch.empty((n_clips, nk, 2), dtype=torch.float32, device="cuda")
    gpu.stage_spectrum_dev(d_pcm.data_ptr(), n, n_clips, d_x.data_ptr())
    torch.cuda.synchronize()
    x_gpu = d_x.cpu().numpy()
    x_ref = np.stack([plan.spectrum(c) for c in clips])
    assert np.array_equal(x_gpu, x_ref), _report("spectrum", x_gpu, x_ref)
    # chirp-z bands -> |c_j[3c]|
    d_mag = torch.empty((n_clips, 121, plan.c), dtype=torch.float32, device="cuda")
    gpu.stage_cqmag_dev(d_x.data_ptr(), n, n_clips, d_mag.data_ptr())
    torch.cuda.synchronize()
    mag_gpu = d_mag.cpu().numpy()
    mag_ref = np.stack([plan.cqmag(x) for x in x_ref])
    assert np.array_equal(mag_gpu, mag_ref), _report("cqmag", mag_gpu, mag_ref)
    # north_star tolerance against the float64 definition: 1e-4 relative (to each band's maximum)
    from oracle import nsgt_f64
    m64 = nsgt_f64.cq_magnitudes(clips[0])
    rel = np.abs(mag_gpu[0] - m64).max(axis=1) / m64.max(axis=1)
    assert rel.max() < 1e-4, f"CQ magnitudes vs float64 definition: {rel.max():.3e}"
    # dB
    d_db = torch.empty_like(d_mag)
    gpu.stage_db_dev(d_mag.data_ptr(), n_clips, plan.c, d_db.data_ptr())
    torch.cuda.synchronize()
    db_gpu = d_db.cpu().numpy()
    db_ref = np.stack([oracle.db(m) for m in mag_ref])
    assert bits_equal(db_gpu, db_ref), _report("db", db_gpu, db_ref)
    # the same spectrogram as extraction's front end produces it (dB terms written by the chirp-z
    # kernel, reference level applied afterwards)
    d_db2 = torch.empty_like(d_mag)
    gpu.stage_spectrogram_dev(d_pcm.data_ptr(), n, n_clips, d_db2.data_ptr())
    torch.cuda.synchronize()
    assert bits_equal(d_db2.cpu().numpy(), db_ref), _report("spectrogram", d_db2.cpu().numpy(), db_ref)
    # projection (f32 MFMA) against the fmaf chain
    d_proj = torch.empty((n_clips, 64, plan.n_frames), dtype=torch.float32, device="cuda")
    gpu.stage_project_dev(d_db.data_ptr(), n_clips, plan.c, d_proj.data_ptr())
    torch.cuda.synchronize()
    pr_gpu = d_proj.cpu().numpy()
    pr_ref = np.stack([oracle.project(filters, s) for s in db_ref])
    assert bits_equal(pr_gpu, pr_ref), _report("project", pr_gpu, pr_ref)
    # delta + pack
    d_hp = torch.empty((n_clips, plan.n_hp), dtype=torch.int64, device="cuda")
    gpu.stage_pack_dev(d_proj.data_ptr(), n_clips, plan.n_frames, d_hp.data_ptr())
    torch.cuda.synchronize()
    hp_gpu = d_hp.cpu().numpy().view(np.uint64)
    hp_ref = np.stack([oracle.pack(p) for p in pr_ref])
    assert np.array_equal(hp_gpu, hp_ref)
    # whole chain, host entry point
    hp_all = gpu.extract(clips)
    assert np.array_equal(hp_all, np.stack([plan.extract(filters, c) for c in clips]))
    # (extraction's default projection is the fixed-point one: against the f32 chain's hashprints above it may differ
    # where a difference of two projections lies within rounding of zero)
    flips = sum(bin(int(v)).count("1") for v in (hp_all ^ hp_ref).ravel())
    assert flips <= 1e-5 * hp_all.size * 64, flips


@pytest.mark.parametrize("seconds", [2.0, 4.2, 7.0, 10.0, 12.5, 20.0, 45.0, 49.0, 60.0, 100.0, 180.0])
def test_extract_other_lengths(gpu, oracle, filters, seconds):
    """other clip lengths (different n1, chirp-z classes from 64 to 16384 points in LDS and, from 60 s on,
    lengths up to 98304 points through global memory, the run-time group sequence when n2 is not 6300):
    hashprints and the dB spectrogram stay bit-exact"""
    torch = pytest.importorskip("torch")
    clips = np.stack([synth.gen_clip(900 + i, seconds) for i in range(2 if seconds < 60 else 1)])
    n = clips.shape[1]
    plan = oracle.Plan(n)
    assert np.array_equal(gpu.extract(clips), np.stack([plan.extract(filters, c) for c in clips]))
    d_pcm = _dev(torch, clips)
    d_db = torch.empty((len(clips), 121, plan.c), dtype=torch.float32, device="cuda")
    gpu.stage_spectrogram_dev(d_pcm.data_ptr(), n, len(clips), d_db.data_ptr())
    torch.cuda.synchronize()
    want = np.stack([oracle.db(plan.cqmag(plan.spectrum(c))) for c in clips])
    assert bits_equal(d_db.cpu().numpy(), want)


@pytest.mark.parametrize("seconds", [1.0, 3.0, 30.0])
def test_mel_front_end(gpu, oracle, seconds):
    """f3: the Mel front-end (mel.h:34-104): frames kept, their order and every dB value identical to the
    oracle's (the STFT rides the forward transform's row kernel, the filterbank runs on f32 MFMA)"""
    mel = oracle.Mel()
    clips = np.stack([synth.gen_clip(950 + i, seconds) for i in range(3)])
    clips[1, 5000:30000] = 0                               # silent frames in the middle of a clip
    clips[2, -9000:] = 0                                   # ... and at its end
    if seconds == 1.0:
        clips[0, :] = 0                                    # an entirely silent clip: no column at all
    got = gpu.mel_spectrogram(clips)
    for c, g in zip(clips, got):
        want = mel.spectrogram(c)
        assert g.shape == want.shape
        assert bits_equal(g, want), _report("mel", g, want)
    odd = gpu.mel_spectrogram(clips[1, :clips.shape[1] - 12345])   # any length, not only 7-smooth ones
    assert bits_equal(odd[0], mel.spectrogram(clips[1, :clips.shape[1] - 12345]))


@pytest.mark.parametrize("n,shift", [(1323000, 0), (1323000, 1), (1323000, 2), (110250, 0), (110250, 1), (99225, 0), (99225, 1),
                                     (2646000, 0)])
def test_column_stage_load_widths_and_sample_extremes(gpu, torch_cuda, oracle, n, shift):
    """S6's column stage reads the PCM where the caller put it: rows of n2 samples in 8-, 4- or 2-byte loads by what n2 and
    the pointer's alignment allow (n2 = 6300, 5250, 6615; the buffer shifted by one and two samples), two chunks of k1 at
    60 s (n1 = 420), and the ends of the int16 range through the sample digits (x = 256 hi + lo + 128): forward bins
    bit-identical to the oracle's exact integer sums"""
    torch = torch_cuda
    rng = np.random.default_rng(n + shift)
    clips = np.stack([synth.gen_clip(970 + i, n / 44100.0)[:n] for i in range(2)])
    clips[0, ::97] = 32767
    clips[0, 5::89] = -32768
    clips[1, : n // 3] = rng.integers(-32768, 32768, n // 3).astype(np.int16)   # full-scale noise: every digit value
    plan = oracle.Plan(n)
    buf = torch.zeros(2 * n + 8, dtype=torch.int16, device="cuda")
    buf[shift: shift + 2 * n] = torch.from_numpy(clips.reshape(-1)).cuda()
    d_x = torch.empty((2, plan.kmax - plan.kmin, 2), dtype=torch.float32, device="cuda")
    gpu.stage_spectrum_dev(buf.data_ptr() + 2 * shift, n, 2, d_x.data_ptr())
    torch.cuda.synchronize()
    got = d_x.cpu().numpy()
    for i in range(2):
        assert bits_equal(got[i], plan.spectrum(clips[i])), (n, shift, i)


def test_clip_too_short_is_an_error(gpu, filters):
    """below 100 spectrogram columns there is no hashprint (the reference would resize a matrix to a
    negative width, hashprint_handle.h:118); the library says so"""
    with pytest.raises(hpfw_amd.HpfwError) as e:
        gpu.extract(np.zeros((1, 44100), np.int16))
    assert "too short" in str(e.value)


def test_extract_batches_and_order(gpu, oracle, filters):
    """more clips than one internal pass, odd batch size: per-clip results do not depend on batching"""
    clips = np.stack([synth.gen_clip(200 + i, 3.0) for i in range(11)])
    plan = oracle.Plan(clips.shape[1])
    want = plan.extract_batch(filters, clips, n_threads=4)
    gpu.set_batch(4)
    got4 = gpu.extract(clips)
    gpu.set_batch(0)
    got = gpu.extract(clips)
    assert np.array_equal(got4, want) and np.array_equal(got, want)


def test_projection_edge_values(gpu, torch_cuda, oracle, filters):
    """exact zeros (>= 0 -> bit 1), the -80 dB floor everywhere, and a tile that is not a multiple of 256"""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    for c in (100, 275, 276, 531):
        s = rng.uniform(-80, 0, (2, 121, c)).astype(np.float32)
        s[1] = -80.0
        s[0, :, : c // 3] = 0.0
        d_s = _dev(torch, s)
        nf = c - 19
        d_proj = torch.empty((2, 64, nf), dtype=torch.float32, device="cuda")
        gpu.stage_project_dev(d_s.data_ptr(), 2, c, d_proj.data_ptr())
        d_hp = torch.empty((2, nf - 80), dtype=torch.int64, device="cuda")
        gpu.stage_pack_dev(d_proj.data_ptr(), 2, nf, d_hp.data_ptr())
        torch.cuda.synchronize()
        pr_ref = np.stack([oracle.project(filters, x) for x in s])
        assert bits_equal(d_proj.cpu().numpy(), pr_ref), _report("project", d_proj.cpu().numpy(), pr_ref)
        hp_ref = np.stack([oracle.pack(p) for p in pr_ref])
        assert np.array_equal(d_hp.cpu().numpy().view(np.uint64), hp_ref)
    # a constant spectrogram gives delta == 0 everywhere -> all 64 bits set (hashprint_handle.h:121)
    assert (hp_ref[1] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()


def _ragged(rng, lens):
    hp = [rng.integers(0, 2 ** 64, size=n, dtype=np.uint64) for n in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return (np.concatenate(hp) if hp else np.zeros(0, np.uint64)), off


def test_search_matches_oracle(gpu, oracle, scan_path):
    rng = np.random.default_rng(11)
    db_lens = [2320, 305, 40, 1000, 2320, 1, 700, 305, 2320, 64, 333, 2000, 5]
    db, db_off = _ragged(rng, db_lens)
    qs = []
    # planted (noisy) slices, an exact slice, a query longer than some clips, a 1-long query, random
    for c, o, k, flips in [(0, 100, 305, 5), (4, 2015, 305, 0), (3, 0, 1000, 9), (6, 10, 64, 3), (11, 7, 1, 0)]:
        seg = db[db_off[c] + o: db_off[c] + o + k].copy()
        for _ in range(flips):
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        qs.append(seg)
    qs.append(rng.integers(0, 2 ** 64, size=305, dtype=np.uint64))
    qs += [rng.integers(0, 2 ** 64, size=int(n), dtype=np.uint64) for n in rng.integers(1, 400, 9)]
    q_off = np.concatenate([[0], np.cumsum([q.size for q in qs])]).astype(np.int64)
    q = np.concatenate(qs)
    gpu.index_clear()
    gpu.index_add(db, db_off)
    assert gpu.index_size() == len(db_lens)
    for k in (1, 10, 13, 20):
        got = gpu.search_topk(q, q_off, k)
        want = oracle.search_topk(db, db_off, q, q_off, k, n_threads=4)
        assert np.array_equal(got, want), f"top-{k} differs"
    # storage.h:37-39 clamps k to the clip length, so the 1- and 5-word clips score at most 64 / 320
    # and outrank a noisy true match; the planted slices are still found at their offsets
    hit0 = got[0][got[0]["clip"] == 0][0]
    assert hit0["offset"] == 100 and hit0["dist"] <= 5 * 305
    assert got[1, 0]["clip"] == 4 and got[1, 0]["offset"] == 2015 and got[1, 0]["dist"] == 0


def test_search_ties_and_duplicates(gpu, oracle, scan_path):
    """identical clips and repeated content: first strict minimum in database order and the
    smallest offset win (storage.h:50,56)"""
    rng = np.random.default_rng(12)
    base = rng.integers(0, 2 ** 64, size=400, dtype=np.uint64)
    rep = np.concatenate([base[:100], base[:100], base[:100]])   # the same 100 words three times
    clips = [base, rep, base.copy(), np.zeros(300, np.uint64), np.zeros(300, np.uint64)]
    db = np.concatenate(clips)
    db_off = np.concatenate([[0], np.cumsum([c.size for c in clips])]).astype(np.int64)
    qs = [base[:100], np.zeros(50, np.uint64), base[50:150]]
    q = np.concatenate(qs)
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    gpu.index_clear()
    gpu.index_add(db, db_off)
    got = gpu.search_topk(q, q_off, 5)
    want = oracle.search_topk(db, db_off, q, q_off, 5)
    assert np.array_equal(got, want)
    assert [int(x) for x in got[0]["clip"][:3]] == [0, 1, 2] and (got[0]["dist"][:3] == 0).all()
    assert (got[0]["offset"][:3] == 0).all()            # first offset of the repeated block
    assert [int(x) for x in got[1]["clip"][:2]] == [3, 4]


def test_search_many_ragged_queries(gpu, oracle, scan_path):
    """several groups of 32 queries with mixed lengths (empty, 1, longer than some clips), clips from 1
    to 3000 hashprints (more than one 1024-offset chunk): both kernels equal the oracle"""
    rng = np.random.default_rng(14)
    db_lens = [3000, 2320, 1, 2, 64, 1024, 1025, 1329, 2048, 2049, 31, 33, 500, 2320]
    db, db_off = _ragged(rng, db_lens)
    lens = [0, 1, 2, 31, 32, 33, 64, 305, 304, 700, 1500, 2400] + [int(x) for x in rng.integers(1, 420, 66)]
    qs = []
    for i, k in enumerate(lens):
        c = i % len(db_lens)
        if k and k <= db_lens[c] and i % 3:
            o = int(rng.integers(0, db_lens[c] - k + 1))
            seg = db[db_off[c] + o: db_off[c] + o + k].copy()
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        else:
            seg = rng.integers(0, 2 ** 64, size=k, dtype=np.uint64)
        qs.append(seg)
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    q = np.concatenate(qs)
    gpu.index_clear()
    gpu.index_add(db, db_off)
    got = gpu.search_topk(q, q_off, 7)
    want = oracle.search_topk(db, db_off, q, q_off, 7, n_threads=4)
    assert np.array_equal(got, want)
    assert (got[0]["clip"] == 0xFFFFFFFF).all()          # the empty query matches nothing


def test_search_large_index_few_queries(gpu, oracle):
    """20 000 clips and 5 queries: the one-query scan kernel and the two-step top-k (64 slices per query,
    then a merge), duplicates included -- the same hits as the oracle, in the same order"""
    rng = np.random.default_rng(17)
    n_clips, n = 20000, 70
    db = rng.integers(0, 2 ** 64, size=(n_clips, n), dtype=np.uint64)
    db[15000] = db[123]                                   # identical clips far apart: ties by clip id
    db[19999] = db[123]
    db_off = np.arange(n_clips + 1, dtype=np.int64) * n
    qs = [db[123, 3:67].copy(), db[7777, 0:64] ^ np.uint64(5), db[19998, 6:70].copy(),
          rng.integers(0, 2 ** 64, size=64, dtype=np.uint64), db[0, 1:41].copy()]
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    q = np.concatenate(qs)
    gpu.index_clear()
    gpu.index_add(db.ravel(), db_off)
    for k in (1, 10, 12):
        got = gpu.search_topk(q, q_off, k)
        assert np.array_equal(got, oracle.search_topk(db.ravel(), db_off, q, q_off, k, n_threads=4))
    assert [int(x) for x in got[0]["clip"][:3]] == [123, 15000, 19999] and (got[0]["dist"][:3] == 0).all()


def test_search_query_longer_than_lds_window(gpu, oracle):
    """a query of 6000 hashprints does not fit the matrix-core kernel's LDS window: the library
    takes the xor/popcount kernel by itself and the answer is the same"""
    rng = np.random.default_rng(15)
    db, db_off = _ragged(rng, [7000, 100, 6500])
    q = np.concatenate([db[db_off[0] + 300: db_off[0] + 6300], rng.integers(0, 2 ** 64, size=40, dtype=np.uint64)])
    q_off = np.array([0, 6000, 6040], np.int64)
    gpu.index_clear()
    gpu.index_add(db, db_off)
    got = gpu.search_topk(q, q_off, 3)
    assert np.array_equal(got, oracle.search_topk(db, db_off, q, q_off, 3, n_threads=4))
    assert got[0, 0]["clip"] == 0 and got[0, 0]["offset"] == 300 and got[0, 0]["dist"] == 0


def test_voting_search_matches_oracle(gpu, oracle):
    """row f4: AnnStorage::find with exact neighbours -- the 5 nearest 64-hashprint windows of every query
    position (keys identical to the oracle's brute force, ties by database position) and the vote"""
    rng = np.random.default_rng(16)
    db_lens = [400, 64, 63, 1500, 2320, 100, 1100]
    db, db_off = _ragged(rng, db_lens)
    db[db_off[5]:db_off[6]] = db[db_off[0] + 50: db_off[0] + 150]          # clip 5 repeats part of clip 0: ties
    qs = []
    for c, o, k, flips in [(3, 700, 200, 6), (4, 2000, 305, 12), (0, 50, 100, 0), (6, 10, 64, 3)]:
        seg = db[db_off[c] + o: db_off[c] + o + k].copy()
        for _ in range(flips):
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        qs.append(seg)
    qs.append(rng.integers(0, 2 ** 64, size=90, dtype=np.uint64))            # matches nothing in particular
    qs.append(rng.integers(0, 2 ** 64, size=40, dtype=np.uint64))            # shorter than a window: no vote
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    q = np.concatenate(qs)
    gpu.index_clear()
    gpu.index_add(db, db_off)
    keys = gpu.knn_windows(q, q_off)
    want_keys = np.concatenate([oracle.knn_windows(db, db_off, x) for x in qs if x.size >= 64])
    assert np.array_equal(keys, want_keys)
    got = gpu.search_votes(q, q_off)
    for i, x in enumerate(qs):
        want = oracle.search_votes(db, db_off, x)
        if want["clip"] < 0:
            assert got[i]["clip"] == 0xFFFFFFFF and got[i]["cnt"] == 0
        else:
            assert (int(got[i]["clip"]), int(got[i]["offset"])) == (int(want["clip"]), int(want["offset"]))
            assert got[i]["cnt"] == want["cnt"]
    assert (int(got[0]["clip"]), int(got[0]["offset"])) == (3, -700)
    assert (int(got[1]["clip"]), int(got[1]["offset"])) == (4, -2000)
    assert int(got[2]["clip"]) == 0 and int(got[2]["offset"]) == -50        # clip 0 comes before its copy, clip 5
    assert got[5]["clip"] == 0xFFFFFFFF


def test_search_empty_and_small_index(gpu, oracle, scan_path):
    rng = np.random.default_rng(13)
    q = rng.integers(0, 2 ** 64, size=30, dtype=np.uint64)
    q_off = np.array([0, 30], np.int64)
    gpu.index_clear()
    got = gpu.search_topk(q, q_off, 4)
    assert (got["clip"] == 0xFFFFFFFF).all() and (got["dist"] == 0xFFFFFFFF).all()
    db, db_off = _ragged(rng, [50, 60])
    gpu.index_add(db, db_off)
    got = gpu.search_topk(q, q_off, 4)                   # k larger than the index
    want = oracle.search_topk(db, db_off, q, q_off, 4)
    assert np.array_equal(got, want)
    assert (got[0]["clip"][2:] == 0xFFFFFFFF).all()
    # incremental add keeps clip ids in insertion order
    db2, db_off2 = _ragged(rng, [70])
    gpu.index_add(db2, db_off2)
    got = gpu.search_topk(db2[5:35], q_off, 1)
    assert got[0, 0]["clip"] == 2 and got[0, 0]["offset"] == 5 and got[0, 0]["dist"] == 0


def test_full_size_properties(gpu, torch_cuda, oracle, filters):
    """BASELINE-size clips (30 s) and queries (5 s): sampled clips bit-identical to the oracle, and
    the round trip index -> noisy 5 s query -> search finds the right clip at the expected offset."""
    n_clips = 12
    clips = [synth.gen_clip(300 + i, 30.0) for i in range(n_clips)]
    pcm = np.stack(clips)
    hp = gpu.extract(pcm)
    assert hp.shape == (n_clips, 2320)
    plan = oracle.Plan(pcm.shape[1])
    for i in (0, 7, 11):
        assert np.array_equal(hp[i], plan.extract(filters, clips[i]))
    off = np.arange(0, (n_clips + 1) * 2320, 2320, dtype=np.int64)
    gpu.index_clear()
    gpu.index_add(hp, off)
    queries = [synth.gen_query(clips, q) for q in range(8)]
    qpcm = np.stack([x[0] for x in queries])
    qhp = gpu.extract(qpcm)
    assert qhp.shape[1] == 304
    qplan = oracle.Plan(qpcm.shape[1])
    assert np.array_equal(qhp[3], qplan.extract(filters, qpcm[3]))
    q_off = np.arange(0, 9 * 304, 304, dtype=np.int64)
    hits = gpu.search_topk(qhp, q_off, 3)
    want = oracle.search_topk(hp.ravel(), off, qhp.ravel(), q_off, 3, n_threads=8)
    assert np.array_equal(hits, want)
    hop = 1323000 / 7255 * 3                             # samples per spectrogram column
    for qi, (_, ci, start) in enumerate(queries):
        assert hits[qi, 0]["clip"] == ci
        assert abs(hits[qi, 0]["offset"] - start / hop) <= 2


def test_golden_vectors_on_gpu(gpu, torch_cuda):
    """the committed fixtures (tests/golden, generated by make_golden.py) through the HIP path"""
    import os
    import sys
    import hpfw_amd
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import gen
    g = np.load(os.path.join(here, "golden", "extract.npz"))
    h = hpfw_amd.Gpu(0)
    h.set_filters(gen.golden_filters())
    two = np.stack([gen.golden_pcm(110250, 1), gen.golden_pcm(110250, 2)])
    hp = h.extract(two)
    assert np.array_equal(hp[0], g["a_hp"]) and np.array_equal(hp[1], g["b_hp"])
    assert np.array_equal(h.extract(gen.golden_pcm(88200, 3))[0], g["c_hp"])
    torch = torch_cuda
    d_pcm = _dev(torch, two)
    geo = g["a_geometry"]
    nk, c = int(geo[3] - geo[2]), int(geo[5])
    d_x = torch.empty((2, nk, 2), dtype=torch.float32, device="cuda")
    h.stage_spectrum_dev(d_pcm.data_ptr(), 110250, 2, d_x.data_ptr())
    d_mag = torch.empty((2, 121, c), dtype=torch.float32, device="cuda")
    h.stage_cqmag_dev(d_x.data_ptr(), 110250, 2, d_mag.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_x.cpu().numpy()[1][::97], g["b_x_every97"])
    mag = d_mag.cpu().numpy()
    assert np.array_equal(mag[0][::4, ::8], g["a_mag_f32_every8"])
    m64 = g["a_mag_f64_every8"]
    assert (np.abs(mag[0][::4, ::8] - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-4   # north_star tolerance
    s = np.load(os.path.join(here, "golden", "search.npz"))
    h.index_add(s["db"], s["db_off"])
    assert np.array_equal(h.search_topk(s["q"], s["q_off"], 5), s["top5"])
    h.close()


def test_plan_cache_eviction(torch_cuda, oracle, filters, monkeypatch):
    """the per-length tables are cached least-recently-used under HPFW_PLAN_CACHE_GB: with room for one
    plan only, alternating lengths rebuild the tables each time and the hashprints do not change"""
    monkeypatch.setenv("HPFW_PLAN_CACHE_GB", "0.0005")   # 0.5 MB: below any single plan
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    clips = {sec: synth.gen_clip(77 + i, sec) for i, sec in enumerate((2.0, 3.0, 4.0))}
    # two lengths with a prime factor above 7 as well: their tables are generated on the device, in memory that comes from
    # the handle's pool and goes back to it at every eviction
    clips["3 s - 1"] = synth.gen_clip(81, 3.0)[:-1]
    clips["2 s + 5"] = np.concatenate([synth.gen_clip(82, 2.0), synth.gen_clip(83, 2.0)[:5]])
    want = {sec: oracle.Plan(c.size).extract(filters, c) for sec, c in clips.items()}
    for sec in (2.0, "3 s - 1", 3.0, 2.0, "2 s + 5", 4.0, "3 s - 1", 3.0, "2 s + 5", 2.0):
        assert np.array_equal(g.extract(clips[sec])[0], want[sec]), sec
    g.close()


def test_extreme_signals(torch_cuda, oracle, filters):
    """silence, DC at full scale, the Nyquist square wave at full scale, a single impulse, white noise at
    full scale: hashprints and dB spectrograms equal the oracle's bit for bit (the dB floor, the clip
    maximum and the largest magnitudes the transform can meet)"""
    torch = torch_cuda
    n = 3 * 44100
    rng = np.random.default_rng(5)
    sig = np.zeros((6, n), np.int16)
    sig[1, :] = -32768
    sig[2, 0::2], sig[2, 1::2] = 32767, -32768
    sig[3, n // 2] = 32767
    sig[4, :] = rng.integers(-32768, 32768, n).astype(np.int16)
    sig[5, :10] = 1                                            # next to silence: magnitudes near the floor
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    plan = oracle.Plan(n)
    hp = g.extract(sig)
    c = int(g.geometry(n).c)
    d_db = torch.empty((6, 121, c), dtype=torch.float32, device="cuda")
    g.stage_spectrogram_dev(_dev(torch, sig).data_ptr(), n, 6, d_db.data_ptr())
    torch.cuda.synchronize()
    db = d_db.cpu().numpy()
    for i in range(6):
        assert np.array_equal(hp[i], plan.extract(filters, sig[i])), i
        assert bits_equal(db[i], oracle.db(plan.cqmag(plan.spectrum(sig[i])))), i
    g.close()


@pytest.mark.parametrize("n", [1323001, 352799, 99991, 220499, 54254, 1764001])
def test_arbitrary_clip_lengths(gpu, torch_cuda, oracle, filters, n):
    """clip lengths with a prime factor above 7 (1323001 = 11 * 120273, 13 | 352799, 99991 prime, ...; 1764001: 40 s,
    two row tiles of the two-stage column transform): the
    reference transforms the file's exact sample count (cqt.h:54-55) and so does the chirp-z forward transform
    (k_bluestein.hip): forward bins and dB spectrogram bit-exact against the oracle, |CQ| within
    1e-4 of every band's maximum against the float64 definition at that exact length, hashprints identical"""
    torch = torch_cuda
    from oracle import nsgt_f64
    sec = n / 44100.0 + 0.1
    clips = np.stack([synth.gen_clip(970 + i, sec)[:n] for i in range(2)])
    plan = oracle.Plan(n)
    g = gpu.geometry(n)
    assert (g.n1, g.n2, g.kmin, g.kmax, g.m, g.c, g.n_hp) == (plan.n1, plan.n2, plan.kmin, plan.kmax, plan.m, plan.c, plan.n_hp)
    assert g.n2 == 6300 and g.n1 * g.n2 >= n + (g.kmax - g.kmin) - 1
    nk = plan.kmax - plan.kmin
    d_pcm = _dev(torch, clips)
    d_x = torch.empty((2, nk, 2), dtype=torch.float32, device="cuda")
    gpu.stage_spectrum_dev(d_pcm.data_ptr(), n, 2, d_x.data_ptr())
    torch.cuda.synchronize()
    x_ref = np.stack([plan.spectrum(c) for c in clips])
    assert np.array_equal(d_x.cpu().numpy(), x_ref), _report("spectrum", d_x.cpu().numpy(), x_ref)
    ref64 = np.fft.fft(clips[0].astype(np.float64) / 32768.0)[plan.kmin:plan.kmax]
    assert np.abs((x_ref[0][:, 0] + 1j * x_ref[0][:, 1]) - ref64).max() < 2e-6 * np.abs(ref64).max()
    d_mag = torch.empty((2, 121, plan.c), dtype=torch.float32, device="cuda")
    gpu.stage_cqmag_dev(d_x.data_ptr(), n, 2, d_mag.data_ptr())
    d_db = torch.empty_like(d_mag)
    gpu.stage_spectrogram_dev(d_pcm.data_ptr(), n, 2, d_db.data_ptr())
    torch.cuda.synchronize()
    mag = d_mag.cpu().numpy()
    mag_ref = np.stack([plan.cqmag(x) for x in x_ref])
    assert np.array_equal(mag, mag_ref)
    m64 = nsgt_f64.cq_magnitudes(clips[0])
    assert (np.abs(mag[0] - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-4
    assert bits_equal(d_db.cpu().numpy(), np.stack([oracle.db(m) for m in mag_ref]))
    if plan.n_hp > 0:
        assert np.array_equal(gpu.extract(clips), np.stack([plan.extract(filters, c) for c in clips]))


@pytest.mark.parametrize("n", [99991, 352799, 1323001])
def test_chirpz_tables_generated_on_device(gpu, oracle, n):
    """S15: chirp, T_L, Bhat and w[k] / L of a clip length are generated on the device (double arithmetic with an own
    cosine / sine, S2b, then one forward transform of the lags): each table identical to the oracle's, bit for bit"""
    plan = oracle.Plan(n)
    for which, name in enumerate(("w", "T_L", "Bhat", "w[k]/L")):
        got, want = gpu.chirpz_table(n, which), plan.chirpz_table(which)
        assert got.shape == want.shape, name
        bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
        assert bad.size == 0, (name, bad.size, bad[:5], got[bad[:5]], want[bad[:5]])
    with pytest.raises(hpfw_amd.HpfwError):
        gpu.chirpz_table(132300, 0)                               # a 7-smooth length has none


@pytest.mark.parametrize("n", [1323000, 1323001, 220500, 99991, 54254])
def test_constant_q_windows_generated_on_device(gpu, oracle, n):
    """S5: the window table of a clip length -- 121 bands of hann_Lg[i] e^{+i pi 3 i^2 / M} / (M P) -- is generated on the
    device (k_cq_tables.hip; double arithmetic with the own cosine / sine of S2b, csrc/trig_d.h), for 7-smooth lengths and
    the others alike: identical to the oracle's table, bit for bit.  (hpfw_gpu_plan_checksum_ex, tests/test_library.py,
    pins the host's restatement of the same text to the oracle without a GPU.)"""
    got, want = gpu.chirpz_table(n, 4), oracle.Plan(n).chirpz_table(4)
    assert got.shape == want.shape
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, (bad.size, bad[:5], got[bad[:5]], want[bad[:5]])


def test_three_minute_clip_of_a_non_smooth_length(gpu, torch_cuda, oracle, filters):
    """3 minutes + 1 sample (n1 = 1392: six row tiles of the two-stage column transform; constant-Q classes above the
    LDS, k_cq_big.hip): the device-generated tables, the forward bins and the hashprints, bit for bit"""
    torch = torch_cuda
    n = 180 * 44100 + 1
    plan = oracle.Plan(n)
    assert gpu.geometry(n).n1 == plan.n1 == 1392
    for which in range(4):
        assert np.array_equal(gpu.chirpz_table(n, which).view(np.uint64), plan.chirpz_table(which).view(np.uint64)), which
    clip = synth.gen_clip(4242, 180.1)[:n]
    d_pcm = _dev(torch, clip[None])
    d_x = torch.empty((1, plan.kmax - plan.kmin, 2), dtype=torch.float32, device="cuda")
    gpu.stage_spectrum_dev(d_pcm.data_ptr(), n, 1, d_x.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_x.cpu().numpy()[0], plan.spectrum(clip))
    assert np.array_equal(gpu.extract(clip[None]), plan.extract(filters, clip)[None])


def test_chirpz_path_agrees_with_mixed_radix_path(oracle, filters, monkeypatch):
    """a 7-smooth length through both forward transforms (HPFW_FORCE_BLUESTEIN=1): each bit-exact against its
    own oracle twin; the two spectrograms agree to float rounding and the hashprints differ in a handful of
    bits at most (two evaluations of the same transform)"""
    clips = np.stack([synth.gen_clip(980 + i, 5.0) for i in range(3)])
    n = clips.shape[1]
    a = hpfw_amd.Gpu(0)
    a.set_filters(filters)
    hp_a = a.extract(clips)
    a.close()
    monkeypatch.setenv("HPFW_FORCE_BLUESTEIN", "1")
    b = hpfw_amd.Gpu(0)                                     # the switch is read when the plan of a length is built
    b.set_filters(filters)
    assert b.geometry(n).n2 == 6300 and b.geometry(n).n1 * 6300 > n
    hp_b = b.extract(clips)
    b.close()
    plan_b = oracle.Plan(n, force_bluestein=True)
    assert np.array_equal(hp_b, np.stack([plan_b.extract(filters, c) for c in clips]))
    assert np.array_equal(hp_a, np.stack([oracle.Plan(n).extract(filters, c) for c in clips]))
    flipped = sum(bin(int(v)).count("1") for v in (hp_a ^ hp_b).ravel())
    assert flipped <= 1e-3 * hp_a.size * 64, flipped


def test_many_clips_of_a_non_smooth_length(gpu, oracle, filters):
    """more clips than one pass of the chirp-z path and an odd count: per-clip results do not depend on batching"""
    n = 3 * 44100 + 7
    clips = np.stack([synth.gen_clip(990 + i, 3.1)[:n] for i in range(9)])
    want = oracle.Plan(n).extract_batch(filters, clips, n_threads=8)
    gpu.set_batch(4)
    got4 = gpu.extract(clips)
    gpu.set_batch(0)
    assert np.array_equal(got4, want) and np.array_equal(gpu.extract(clips), want)


def test_combiner_configuration_16_bit_hashprints(gpu, torch_cuda, oracle):
    """f3: HashprintHandle<uint16_t, MelSpectrogram<>, 32, 50> (combiner.h:12; hashprint_handle.h:50-64 with
    33 rows, frames of 1056 values, 16 filters): projection bit-exact against the fmaf chain, 16-bit hashprints
    identical, from given spectrograms of ragged widths and end to end from PCM through the Mel front end"""
    torch = torch_cuda
    cfg = hpfw_amd.COMBINER_CONFIG
    rows, ctx, lag, bits = cfg
    rng = np.random.default_rng(21)
    filt = np.linalg.qr(rng.standard_normal((rows * ctx, bits)))[0].astype(np.float32)   # [k][r] = column-major [bits][k]
    gpu.cfg_set_filters(cfg, filt)
    stride = 700
    cols = np.array([700, 82, 81, 400, 33, 0], np.int32)          # 82: one hashprint; 81: none; fewer than a frame: none
    s = rng.uniform(-80, 0, (len(cols), rows, stride)).astype(np.float32)
    s[3, :, :200] = -80.0                                         # constant stretch: exact zeros in the deltas
    nf = stride - ctx + 1
    d_s, d_cols = _dev(torch, s), _dev(torch, cols)
    d_hp = torch.zeros((len(cols), nf - lag), dtype=torch.int16, device="cuda")
    d_proj = torch.zeros((len(cols), bits, nf), dtype=torch.float32, device="cuda")
    gpu.cfg_hashprints_dev(cfg, d_s.data_ptr(), d_cols.data_ptr(), len(cols), stride, d_hp.data_ptr(), nf - lag, d_proj.data_ptr())
    torch.cuda.synchronize()
    hp, proj = d_hp.cpu().numpy().view(np.uint16), d_proj.cpu().numpy()
    for i, c in enumerate(cols):
        want_hp, want_proj = oracle.hashprints_cfg(filt, s[i][:, :c], ctx, lag, bits, return_projection=True)
        n = max(c - ctx + 1, 0)
        assert bits_equal(proj[i][:, :n], want_proj), i
        assert np.array_equal(hp[i][:max(n - lag, 0)], want_hp.astype(np.uint16)), i
    assert (hp[3][:100] == 0xFFFF).all()                          # delta == 0 -> bit set (hashprint_handle.h:121)
    # end to end: PCM -> dB-mel spectrogram (silent frames dropped) -> 16-bit hashprints
    clips = np.stack([synth.gen_clip(960 + i, 6.0) for i in range(3)])
    clips[1, 50000:120000] = 0
    clips[2, :] = 0                                               # silent: no column, no hashprint
    got = gpu.mel_hashprints(clips)
    mel = oracle.Mel()
    for c, g in zip(clips, got):
        sp = mel.spectrogram(c)
        want = oracle.hashprints_cfg(filt, sp, ctx, lag, bits).astype(np.uint16) if sp.shape[1] else np.zeros(0, np.uint16)
        assert np.array_equal(g, want)
    assert got[0].size == oracle.Mel.frames(clips.shape[1]) - 81 and got[2].size == 0 and 0 < got[1].size < got[0].size
    # the same kernels with the live-id arguments <uint64_t, CQT<>, 20, 80> agree with the specialised path
    filt64 = synth.make_filters()
    gpu.cfg_set_filters((121, 20, 80, 64), filt64)
    s64 = rng.uniform(-80, 0, (2, 121, 300)).astype(np.float32)
    d_hp64 = torch.zeros((2, 300 - 99), dtype=torch.int64, device="cuda")
    gpu.cfg_hashprints_dev((121, 20, 80, 64), _dev(torch, s64).data_ptr(), 0, 2, 300, d_hp64.data_ptr(), 300 - 99)
    torch.cuda.synchronize()
    for i in range(2):
        assert np.array_equal(d_hp64.cpu().numpy().view(np.uint64)[i], oracle.pack(oracle.project(filt64, s64[i])))


@pytest.mark.parametrize("conv", [1, 2, 4, 8, 15])
def test_switchable_essentia_conventions(torch_cuda, oracle, filters, conv):
    """the four conventions of essentia's NSGConstantQ that cannot be checked offline (Hann end point, Lg rounding,
    float geometry, inverse-FFT scale; include/hpfw_gpu.h HPFW_CONV_*) are plan-level switches: under every
    setting the GPU path stays bit-identical to the oracle and within 1e-4 of the float64 definition evaluated
    under the same setting -- no kernel knows about them"""
    from oracle import nsgt_f64
    torch = torch_cuda
    clips = np.stack([synth.gen_clip(930 + i, 3.0) for i in range(2)])
    n = clips.shape[1]
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    base = g.extract(clips)
    g.set_conventions(conv)                                   # rebuilds the tables of every length
    plan = oracle.Plan(n, conventions=conv)
    geo = g.geometry(n)
    assert (geo.m, geo.c, geo.kmin, geo.kmax) == (plan.m, plan.c, plan.kmin, plan.kmax)
    assert np.array_equal(g.extract(clips), np.stack([plan.extract(filters, c) for c in clips]))
    nk = plan.kmax - plan.kmin
    d_pcm = _dev(torch, clips)
    d_x = torch.empty((2, nk, 2), dtype=torch.float32, device="cuda")
    g.stage_spectrum_dev(d_pcm.data_ptr(), n, 2, d_x.data_ptr())
    d_mag = torch.empty((2, 121, plan.c), dtype=torch.float32, device="cuda")
    g.stage_cqmag_dev(d_x.data_ptr(), n, 2, d_mag.data_ptr())
    torch.cuda.synchronize()
    mag = d_mag.cpu().numpy()
    m64 = nsgt_f64.cq_magnitudes(clips[0], conv)
    assert mag[0].shape == m64.shape and (np.abs(mag[0] - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-4
    g.set_conventions(0)
    assert np.array_equal(g.extract(clips), base)
    g.close()
