"""Randomised parity run (GPU): extraction at random supported lengths and batch sizes, and ragged searches
through the three scan kernels, all against the oracle.  python3 tools/fuzz_parity.py [rounds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

def _factors(m):
    out, f = [], 2
    while m > 1:
        while m % f == 0:
            out.append(f)
            m //= f
        f += 1
    return out


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
filt = synth.make_filters()
g = hpfw_amd.Gpu(0)
g.set_filters(filt)
bad = 0
t0 = time.time()
for r in range(rounds):
    if r % 2 == 0:   # a 7-smooth length (mixed-radix forward transform), 1.4 s to 75 s
        smooth = [m for m in range(10, 526) if all(p in (2, 3, 5, 7) for p in _factors(m))]
        n = 6300 * int(rng.choice(smooth))
    else:            # any length (almost surely with a prime factor above 7: the chirp-z forward transform), up to
        n = int(rng.integers(60000, 44100 * 75))   # 75 s: one to three row tiles of the two-stage column transform
    nb = int(rng.integers(1, 5))
    g.set_batch(int(rng.integers(1, 4)))
    clips = np.stack([synth.gen_clip(int(rng.integers(1, 1 << 30)), n / 44100.0)[:n] for _ in range(nb)])
    if clips.shape[1] != n:                      # gen_clip rounds the length: pad with noise-free zeros
        clips = np.pad(clips, ((0, 0), (0, n - clips.shape[1])))
    plan = oracle.Plan(n)
    want = np.stack([plan.extract(filt, c) for c in clips])
    got = g.extract(clips)
    ok = np.array_equal(got, want)
    bad += not ok
    print(f"extract round {r}: n={n} ({n / 44100:.2f} s) clips={nb} n1={plan.n1} n2={plan.n2} C={plan.c} "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
g.set_batch(0)
for r in range(rounds):
    n_clips = int(rng.integers(1, 40))
    lens = [int(x) for x in rng.integers(1, 3000, n_clips)]
    db = rng.integers(0, 2 ** 64, size=sum(lens), dtype=np.uint64)
    db_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n_q = int(rng.integers(1, 70))
    qs = []
    for _ in range(n_q):
        k = int(rng.integers(0, 500))
        c = int(rng.integers(0, n_clips))
        if k and k <= lens[c] and rng.random() < 0.6:
            o = int(rng.integers(0, lens[c] - k + 1))
            seg = db[db_off[c] + o: db_off[c] + o + k].copy()
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        else:
            seg = rng.integers(0, 2 ** 64, size=k, dtype=np.uint64)
        qs.append(seg)
    q = np.concatenate(qs) if sum(x.size for x in qs) else np.zeros(0, np.uint64)
    q_off = np.concatenate([[0], np.cumsum([x.size for x in qs])]).astype(np.int64)
    if q.size == 0:
        continue
    g.index_clear()
    g.index_add(db, db_off)
    topk = int(rng.integers(1, 12))
    want = oracle.search_topk(db, db_off, q, q_off, topk, n_threads=8)
    res = []
    for var in ("HPFW_SEARCH_MFMA", "HPFW_SEARCH_SHIFT", "HPFW_SEARCH_POPC", None):
        if var:
            os.environ[var] = "1"
        ok = np.array_equal(g.search_topk(q, q_off, topk), want)
        if var:
            del os.environ[var]
        res.append(ok)
        bad += not ok
    print(f"search round {r}: clips={n_clips} queries={n_q} k={topk} mfma/shift/popc/default={res}", flush=True)
print(f"done in {time.time() - t0:.0f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
