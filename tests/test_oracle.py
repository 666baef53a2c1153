"""CPU tests of the oracle: against an independent float64 statement of the mathematics
(oracle/nsgt_f64.py, numpy FFT), against plain-Python restatements of the integer stages, and
against the committed golden vectors.  The oracle is the checker for the GPU tests; these tests
are what pins it (the reference has no fixtures: DESIGN.md "Oracle", parity unpinned)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen  # noqa: E402
from oracle import nsgt_f64  # noqa: E402


def test_geometry_of_baseline_configs(oracle):
    """30 s -> M 7255, C 2419, 2400 frames, 2320 hashprints; 5 s -> M 1209,
    403 valid columns, 304 hashprints (the uninitialised 404th column of cqt.h:73 is dropped)."""
    p = oracle.Plan(1323000)
    assert (p.m, p.c, p.n_frames, p.n_hp) == (7255, 2419, 2400, 2320)
    assert p.n1 * p.n2 == 1323000 and p.lg.min() == 227 and p.lg.max() == 7255 and int(p.lg.sum()) == 247102
    q = oracle.Plan(220500)
    assert (q.m, q.c, q.n_frames, q.n_hp) == (1209, 403, 384, 304)
    assert q.lg.min() == 96                                  # minimumWindow clamp, cqt.h:58
    posit, lg = nsgt_f64.bands(1323000)
    assert np.array_equal(lg, p.lg) and np.array_equal(posit - lg // 2, p.start)
    r = oracle.Plan(1323001)                                 # 11 * 120273: the chirp-z forward transform (S15)
    assert (r.n2, r.n1) == (6300, 240) and (r.m, r.c) == (7255, 2419)   # n1 = 16 a: two-stage column transform
    with pytest.raises(ValueError):
        oracle.Plan(4410)                                    # bands leave the half spectrum


@pytest.mark.parametrize("n,radix", [(4, [4]), (6, [3, 2]), (35, [7, 5]), (420, [7, 5, 4, 3]),
                                     (14700, [7, 7, 5, 5, 4, 3]), (4096, [4] * 6), (8192, [4] * 6 + [2])])
def test_fft_passes_against_numpy(oracle, n, radix):
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    pos = np.array([oracle.digit_pos(k, n, radix) for k in range(n)])
    ref = np.fft.fft(a.astype(np.complex128))
    got = oracle.fft_dif(a, radix)
    assert np.abs(got[pos] - ref).max() / np.abs(ref).max() < 5e-7
    rev = np.zeros(n, np.complex64)
    rev[pos] = ref.astype(np.complex64)
    back = oracle.fft_idit(rev, radix)
    assert np.abs(back / n - a).max() < 2e-6


def test_twiddles_are_exactly_symmetric(oracle):
    assert oracle.twiddle(0, 8) == (1.0, -0.0)
    assert oracle.twiddle(2, 8) == (-0.0, -1.0) and oracle.twiddle(4, 8) == (-1.0, 0.0)
    for n in (14700, 4096, 90):
        for m in (1, 7, n // 3):
            c, s = oracle.twiddle(m, n)
            c2, s2 = oracle.twiddle(n - m, n)
            assert (c, s) == (c2, -s2)
            assert abs(c - np.cos(2 * np.pi * m / n)) < 6e-8 and abs(s + np.sin(2 * np.pi * m / n)) < 6e-8


def test_log10_spec(oracle):
    rng = np.random.default_rng(3)
    xs = np.concatenate([10.0 ** rng.uniform(-12, 12, 3000), [1.0, 2.0, 0.5, 1e-10, 1.4142135623730951]])
    assert max(abs(oracle.log10(x) - np.log10(x)) for x in xs) < 2e-15


@pytest.mark.parametrize("n", [88200, 110250, 220500])
def test_front_end_against_float64_definition(oracle, n):
    pcm = gen.golden_pcm(n, 5)
    plan = oracle.Plan(n)
    x = plan.spectrum(pcm)
    ref = np.fft.fft(pcm / 32768.0)[plan.kmin:plan.kmax]
    assert np.abs((x[:, 0] + 1j * x[:, 1]) - ref).max() / np.abs(ref).max() < 1e-6
    mag = plan.cqmag(x)
    m64 = nsgt_f64.cq_magnitudes(pcm)
    assert mag.shape == m64.shape
    assert (np.abs(mag - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-5   # north_star asks 1e-4
    s = oracle.db(mag)
    assert s.max() == 0.0 and s.min() >= -80.0
    assert np.abs(s - nsgt_f64.amplitude_to_db(m64)).max() < 0.02


def test_db_floor_and_silence(oracle):
    assert (oracle.db(np.zeros((121, 30), np.float32)) == 0.0).all()   # all at the 1e-10 floor: 0 dB
    m = np.full((121, 30), 1e-7, np.float32)
    m[3, 4] = 1.0
    s = oracle.db(m)
    assert s[3, 4] == 0.0 and (np.delete(s.ravel(), 3 * 30 + 4) == -80.0).all()
    m[0, 0] = 1e-3                                          # -60 dB
    assert abs(oracle.db(m)[0, 0] + 60.0) < 1e-4


def test_projection_and_packing(oracle):
    """layout k = bin * 20 + t (hashprint_handle.h:84-90) and MSB-first bits (:137-142)"""
    rng = np.random.default_rng(8)
    s = rng.uniform(-80, 0, (121, 140)).astype(np.float32)
    f_rows = rng.standard_normal((64, 2420)).astype(np.float32) / 50
    f_cm = np.ascontiguousarray(f_rows.T).ravel()
    pr = oracle.project(f_cm, s)
    assert pr.shape == (64, 121)
    assert np.abs(pr - nsgt_f64.project(f_rows, s)).max() < 2e-2
    one = np.zeros((64, 2420), np.float32)
    one[5, 7 * 20 + 3] = 1.0                                 # filter 5 reads bin 7, context column 3
    assert np.array_equal(oracle.project(np.ascontiguousarray(one.T).ravel(), s)[5], s[7, 3:3 + 121])
    hp = oracle.pack(pr)
    assert hp.shape == (41,) and np.array_equal(hp, nsgt_f64.pack(pr.astype(np.float64)))
    p2 = np.zeros((64, 90), np.float32)
    p2[0, :10] = 1.0                                         # row 0 decreasing over the lag -> MSB
    p2[63, 80:] = -1.0                                       # row 63 -> LSB ... and zeros count as >= 0
    h2 = oracle.pack(p2)
    assert h2[0] == np.uint64(0xFFFFFFFFFFFFFFFF) and h2.shape == (10,)
    p2[0, 80:] = 2.0
    assert oracle.pack(p2)[0] == np.uint64(0x7FFFFFFFFFFFFFFF)


def _popcount(a):
    return sum(bin(int(v)).count("1") for v in a)


def test_match_clip_semantics(oracle):
    """storage.h:33-54: k clamps to the clip length, first strict minimum wins"""
    rng = np.random.default_rng(9)
    r = rng.integers(0, 2 ** 64, size=60, dtype=np.uint64)
    assert oracle.match_clip(r[17:37], r) == (0, 17)
    q = r[17:37].copy()
    q[3] ^= np.uint64(0b1011)
    assert oracle.match_clip(q, r) == (3, 17)
    long_q = np.concatenate([r, rng.integers(0, 2 ** 64, size=10, dtype=np.uint64)])
    assert oracle.match_clip(long_q, r) == (0, 0)            # only the first 60 words are compared
    rep = np.concatenate([r[:10], r[:10], r[:10]])
    assert oracle.match_clip(r[:10], rep) == (0, 0)
    d, off = oracle.match_clip(q, rng.integers(0, 2 ** 64, size=25, dtype=np.uint64))
    assert 0 <= off <= 5 and d > 0


def test_golden_extract(oracle):
    g = np.load(os.path.join(HERE, "golden", "extract.npz"))
    filt = gen.golden_filters()
    for tag in "abc":
        n, seed = int(g[f"{tag}_n"]), int(g[f"{tag}_seed"])
        pcm = gen.golden_pcm(n, seed)
        plan = oracle.Plan(n)
        geo = [plan.n1, plan.n2, plan.kmin, plan.kmax, plan.m, plan.c, plan.n_frames, plan.n_hp]
        assert geo == g[f"{tag}_geometry"].tolist()
        x = plan.spectrum(pcm)
        assert np.array_equal(x[::97], g[f"{tag}_x_every97"])
        mag = plan.cqmag(x)
        assert np.array_equal(mag[::4, ::8], g[f"{tag}_mag_f32_every8"])
        m64 = g[f"{tag}_mag_f64_every8"]
        assert (np.abs(mag[::4, ::8] - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-5
        s = oracle.db(mag)
        assert np.array_equal(s[::4, ::8], g[f"{tag}_db_every8"])
        pr = oracle.project(filt, s)
        assert np.array_equal(pr[::8, ::8], g[f"{tag}_proj_every8"])
        assert np.array_equal(oracle.pack(pr), g[f"{tag}_hp_f32chain"])
        pq = oracle.project_q(filt, s)
        assert np.array_equal(pq[::8, ::8], g[f"{tag}_projq_every8"])
        assert np.array_equal(oracle.pack_q(pq), g[f"{tag}_hp"])
        dq = oracle.delta_q(filt, s)
        assert np.array_equal(dq, pq[:, :-80] - pq[:, 80:]) and np.array_equal(dq[::8, ::8], g[f"{tag}_deltaq_every8"])
        assert oracle.get_projection() == 1 and np.array_equal(plan.extract(filt, pcm), g[f"{tag}_hp"])
    two = np.stack([gen.golden_pcm(110250, 1), gen.golden_pcm(110250, 2)])
    hp = oracle.Plan(110250).extract_batch(filt, two, n_threads=2)
    assert np.array_equal(hp[0], g["a_hp"]) and np.array_equal(hp[1], g["b_hp"])


def test_golden_search(oracle):
    g = np.load(os.path.join(HERE, "golden", "search.npz"))
    for threads in (1, 3):
        top = oracle.search_topk(g["db"], g["db_off"], g["q"], g["q_off"], 5, n_threads=threads)
        assert np.array_equal(top, g["top5"])
    t = g["top5"]
    assert (t[0, 0]["clip"], t[0, 0]["offset"], t[0, 0]["dist"]) == (0, 20, 0)
    assert (t[0, 1]["clip"], t[0, 1]["offset"], t[0, 1]["dist"]) == (4, 20, 0)   # duplicate clip: id order
    # storage.h:37-39 clamps k to the clip length: the 1-word clip 2 and the 5-word clip 6 outrank
    # the planted slice (100 hashprints, two flipped bits each), which comes third at its offset
    assert [int(c) for c in t[1]["clip"][:3]] == [2, 6, 3]
    assert (t[1, 2]["offset"], t[1, 2]["dist"]) == (10, 200)
    top1 = oracle.search_topk(g["db"], g["db_off"], g["q"], g["q_off"], 1)
    assert np.array_equal(top1[:, 0], t[:, 0])


def test_voting_search_oracle_against_numpy(oracle):
    """AnnStorage::find with exact neighbours (annoy_storage.h:41-63): the C restatement against a
    direct numpy evaluation of the same definition"""
    rng = np.random.default_rng(5)
    lens = [150, 64, 30, 260]
    db = rng.integers(0, 2 ** 64, size=sum(lens), dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q = db[off[3] + 40: off[3] + 40 + 100].copy()
    q[::3] ^= np.uint64(0x8001)
    keys = oracle.knn_windows(db, off, q)
    pop = np.array([bin(i).count("1") for i in range(256)], np.uint64)

    def dist(a, b):
        return int(pop[np.bitwise_xor(a, b).view(np.uint8)].sum())

    items = [(c, p) for c in range(len(lens)) for p in range(lens[c] - 63)]
    cnt, best = {}, (-1, 0, np.float32(0))
    for i in range(q.size - 63):
        cand = sorted((dist(q[i:i + 64], db[off[c] + p: off[c] + p + 64]), int(off[c] + p), c, p) for c, p in items)[:5]
        assert [int(k) for k in keys[i]] == [(d << 40) | pos for d, pos, _, _ in cand]
        for d, _, c, p in cand:
            key = (c, i - p)
            cnt[key] = np.float32(np.float64(cnt.get(key, np.float32(0))) + 1.0 / np.float64(np.float32(d + 1)))
            if cnt[key] > best[2]:
                best = (c, i - p, cnt[key])
    got = oracle.vote_windows(keys, off)
    assert (int(got["clip"]), int(got["offset"])) == best[:2] == (3, -40) and got["cnt"] == best[2]
    assert oracle.search_votes(db, off, q[:63])["clip"] == -1      # no window fits


def test_mel_front_end_against_float64(oracle):
    """f3: MelSpectrogram<44100, 33, 4410, 441> (mel.h:34-104) -- the C restatement against the float64
    statement of the same restated essentia algorithms: filterbank, frame count, silent frames, band
    powers, dB spectrogram"""
    from hpfw_amd import synth
    mel = oracle.Mel()
    win, coeff = mel.tables()
    fb = nsgt_f64.mel_filterbank()
    assert coeff.shape == (33, 2206) and np.abs(coeff - fb).max() < 1e-7
    assert np.abs(coeff.sum(axis=1) - 1).max() < 1e-6 and abs(float(win.sum()) - 2.0) < 1e-5
    clip = synth.gen_clip(5, 3.0)
    clip[20000:40000] = 0                                  # a silent stretch: its frames are dropped
    clip[60000:60010] = 1                                  # almost silent: sum of squares 10 <= 473
    p, keep = mel.power(clip)
    p64, keep64 = nsgt_f64.mel_power(clip)
    assert p.shape == (33, oracle.Mel.frames(clip.size)) == (33, (clip.size + 2205 + 440) // 441)
    assert np.array_equal(keep, keep64) and 0 < keep.sum() < keep.size
    assert (np.abs(p - p64).max(axis=1) / p64.max(axis=1)).max() < 1e-5
    s = mel.spectrogram(clip)
    assert s.shape == (33, int(keep.sum())) and s.max() == 0.0 and s.min() >= -80.0
    assert np.abs(s - nsgt_f64.power_to_db(p64[:, keep64])).max() < 1e-3
    assert mel.spectrogram(np.zeros(10000, np.int16)).shape == (33, 0)      # all frames silent


@pytest.mark.parametrize("n", [99991, -132300])
def test_chirpz_tables_against_float64(oracle, n):
    """S15: chirp, T_L, w[k] / L (S2b: own cosine and sine, explicit fma chains) and Bhat (the f32 forward transform of
    the conjugate chirp's lags) against numpy in float64"""
    plan = oracle.Plan(abs(n), force_bluestein=n < 0)
    n = abs(n)
    n1, n2, big_l = plan.n1, plan.n2, plan.n1 * plan.n2
    assert n2 == 6300 and n1 == 16 * -(-(n + (plan.kmax - plan.kmin) - 1) // (16 * 6300))
    j = np.arange(big_l)                                                 # w, the lags: flat index j = n2 k1' + k2'

    def chirp(m):
        m = np.asarray(m, np.int64)
        return np.exp(-1j * np.pi * ((m * m) % (2 * n)) / n)

    w = np.where(j < n, chirp(np.minimum(j, n - 1)), 0)
    assert np.abs(plan.chirpz_table(0) - w).max() < 1e-7
    tl = np.exp(-2j * np.pi * ((np.arange(n1)[:, None] * np.arange(n2)[None, :]) % big_l) / big_l).ravel()
    assert np.abs(plan.chirpz_table(1) - tl).max() < 1e-7
    k = np.arange(plan.kmin, plan.kmax)
    assert np.abs(plan.chirpz_table(3) * big_l - chirp(k)).max() < 1e-7
    b = np.zeros(big_l, np.complex128)
    m = np.arange(plan.kmin - (n - 1), plan.kmax)
    b[m % big_l] = np.conj(chirp(np.abs(m)))
    bhat = np.fft.fft(b).reshape(n2, n1).T.ravel()                         # the table holds frequency q1 + n1 q2 at [q1][q2]
    got = plan.chirpz_table(2).astype(np.complex128)
    assert np.abs(got - bhat).max() < 2e-6 * np.abs(bhat).max()
    assert np.sqrt(np.mean(np.abs(got - bhat) ** 2)) < 3e-7 * np.sqrt(np.mean(np.abs(bhat) ** 2))


@pytest.mark.parametrize("n", [132301, 88211, 99991, 132300])
def test_forward_bins_of_any_length_against_numpy(oracle, n):
    """S15: clip lengths with a prime factor above 7 (11 | 132301, 88211 = 17 * 5189, 99991 prime) and a 7-smooth
    one forced down the same path: the consumed forward bins against numpy's float64 FFT of the exact length"""
    from hpfw_amd import synth
    pcm = synth.gen_clip(31, 3.5)[:n]
    plan = oracle.Plan(n, force_bluestein=True)
    assert plan.n2 == 6300 and plan.n1 * plan.n2 >= n + (plan.kmax - plan.kmin) - 1
    x = plan.spectrum(pcm)
    ref = np.fft.fft(pcm.astype(np.float64) / 32768.0)[plan.kmin:plan.kmax]
    err = np.abs((x[:, 0] + 1j * x[:, 1]) - ref)
    assert err.max() < 2e-6 * np.abs(ref).max()
    if n == 132300:                                          # the mixed-radix transform of the same clip agrees
        y = oracle.Plan(n).spectrum(pcm)
        assert np.abs(x - y).max() < 2e-6 * np.abs(ref).max()


def test_any_length_extraction_against_float64(oracle):
    """a clip of 132301 samples end to end: |CQ| within 1e-4 of every band's maximum against the float64
    definition evaluated at the exact length (BASELINE.json's tolerance)"""
    from hpfw_amd import synth
    n = 132301
    pcm = synth.gen_clip(32, 3.5)[:n]
    plan = oracle.Plan(n)
    mag = plan.cqmag(plan.spectrum(pcm))
    m64 = nsgt_f64.cq_magnitudes(pcm)
    assert mag.shape == m64.shape
    assert (np.abs(mag - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-4


def test_fixed_point_projection_against_numpy(oracle):
    """S9q: quantised factors and exact int64 sums, restated in numpy; the scaled-back values lie within f32 rounding
    of the f32 chain (S9) and of the float64 product"""
    rng = np.random.default_rng(5)
    filt = rng.standard_normal(64 * 2420).astype(np.float32) * 0.02          # column-major [64][2420]
    s = rng.uniform(-80, 0, (121, 140)).astype(np.float32)
    f = filt.reshape(2420, 64).T                                             # [r][k]
    m = np.abs(f).max(axis=1)
    e = 21 - np.floor(np.log2(m)).astype(np.int64)
    fq = np.rint(f.astype(np.float64) * 2.0 ** e[:, None]).astype(np.int64)
    assert np.array_equal(oracle.quantise_filters(filt), fq) and np.abs(fq).max() <= 2 ** 22
    s[3, :7] = [-80.0, 0.0, -79.99999, -1e-6, -40.000004, -80.0, 0.0]          # the ends of the range
    prod = s * np.float32(98304.0)                                           # one f32 rounding, then round-half-even
    assert prod.dtype == np.float32
    u = np.rint(prod.astype(np.float64)).astype(np.int64)
    assert np.array_equal(oracle.quantise_db(s), u) and u.min() >= -80 * 98304 and u.max() <= 0
    nf = s.shape[1] - 19
    frames = np.stack([u[b, t:t + nf] for b in range(121) for t in range(20)])   # [k][n], k = 20 b + t
    want = fq @ frames
    got = oracle.project_q(filt, s)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    hp = oracle.pack_q(got)
    bits = (want[:, :-80] - want[:, 80:]) >= 0
    assert np.array_equal(hp, (bits.astype(np.uint64) << (np.uint64(63) - np.arange(64, dtype=np.uint64))[:, None]).sum(axis=0, dtype=np.uint64))
    # the difference taken first (what the GPU kernel forms): the same integers, and three balanced base-256 digits hold it
    du = u[:, :-80] - u[:, 80:]
    assert np.abs(du).max() <= 80 * 98304 < 127 * 65536 + 127 * 256 + 127
    nhp = nf - 80
    dframes = np.stack([du[b, t:t + nhp] for b in range(121) for t in range(20)])
    dq = oracle.delta_q(filt, s)
    assert np.array_equal(dq, fq @ dframes) and np.array_equal(dq, want[:, :-80] - want[:, 80:])
    back = want / 2.0 ** e[:, None] / 98304.0
    exact = f.astype(np.float64) @ np.stack([s[b, t:t + nf] for b in range(121) for t in range(20)]).astype(np.float64)
    chain = oracle.project(filt, s)
    assert np.abs(back - exact).max() < 2e-4 and np.abs(chain - exact).max() < 2e-3
    assert np.sqrt(np.mean((back - exact) ** 2)) < np.sqrt(np.mean((chain - exact) ** 2))   # closer than the f32 chain
