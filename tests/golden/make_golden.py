#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run here (build container); the fixtures are committed.

No reference run is behind these numbers: the reference ships no fixtures and cannot be built in
this image (DESIGN.md "Oracle": parity unpinned).  The vectors pin (1) the C oracle against
regressions and (2) -- asserted below at generation time -- the C oracle against the independent
float64 numpy statement of the same mathematics (oracle/nsgt_f64.py).  The GPU tests compare the
HIP path with the same files."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import nsgt_f64, oracle  # noqa: E402
import gen  # noqa: E402


def popcount64(a):
    a = a.copy()
    c = np.zeros(a.shape, np.int64)
    for _ in range(64):
        c += (a & np.uint64(1)).astype(np.int64)
        a >>= np.uint64(1)
    return c


def brute_topk(db, db_off, q, q_off, k):
    """plain numpy restatement of storage.h:27-64 + the notebook's top-k, for cross-checking"""
    out = []
    for qi in range(len(q_off) - 1):
        qq = q[q_off[qi]:q_off[qi + 1]]
        rows = []
        for c in range(len(db_off) - 1):
            r = db[db_off[c]:db_off[c + 1]]
            kk = min(len(qq), len(r))
            if kk == 0:
                continue
            best = None
            for off in range(len(r) - kk + 1):
                d = int(popcount64(qq[:kk] ^ r[off:off + kk]).sum())
                if best is None or d < best[0]:
                    best = (d, off)
            rows.append((best[0], c, best[1]))
        rows.sort(key=lambda t: (t[0], t[1]))
        out.append(rows[:k])
    return out


def main():
    oracle.build()
    filt = gen.golden_filters()
    # ---- extraction: two clips of 2.5 s (N = 110250 = 2 3^2 5^3 7^2) and one of 2 s ----
    ext = {}
    for tag, n, seed in (("a", 110250, 1), ("b", 110250, 2), ("c", 88200, 3)):
        pcm = gen.golden_pcm(n, seed)
        plan = oracle.Plan(n)
        x = plan.spectrum(pcm)
        mag = plan.cqmag(x)
        sdb = oracle.db(mag)
        proj = oracle.project(filt, sdb)                               # the f32 chain (S9): kept as a second statement
        hp_f32 = oracle.pack(proj)
        hp = oracle.pack_q(oracle.project_q(filt, sdb))                # fixed point (S9q): what extraction uses
        assert oracle.get_projection() == 1 and np.array_equal(hp, plan.extract(filt, pcm))
        flips = sum(bin(int(v)).count("1") for v in hp ^ hp_f32)
        assert flips <= 1e-4 * hp.size * 64, flips
        # pin the oracle to the float64 definition before trusting it
        xref = np.fft.fft(pcm / 32768.0)[plan.kmin:plan.kmax]
        assert np.abs((x[:, 0] + 1j * x[:, 1]) - xref).max() / np.abs(xref).max() < 1e-6
        m64 = nsgt_f64.cq_magnitudes(pcm)
        assert (np.abs(mag - m64).max(axis=1) / m64.max(axis=1)).max() < 1e-5
        assert np.abs(sdb - nsgt_f64.amplitude_to_db(m64)).max() < 0.02
        p64 = nsgt_f64.project(filt.reshape(2420, 64).T, nsgt_f64.amplitude_to_db(m64))
        assert np.abs(proj - p64).max() < 2e-2
        ext[f"{tag}_n"] = np.int64(n)
        ext[f"{tag}_seed"] = np.int64(seed)
        ext[f"{tag}_geometry"] = np.array([plan.n1, plan.n2, plan.kmin, plan.kmax, plan.m, plan.c,
                                           plan.n_frames, plan.n_hp], np.int64)
        ext[f"{tag}_hp"] = hp
        ext[f"{tag}_hp_f32chain"] = hp_f32
        ext[f"{tag}_projq_every8"] = oracle.project_q(filt, sdb)[::8, ::8].copy()
        ext[f"{tag}_deltaq_every8"] = oracle.delta_q(filt, sdb)[::8, ::8].copy()
        ext[f"{tag}_mag_f32_every8"] = mag[::4, ::8].copy()        # float32, exact oracle bits
        ext[f"{tag}_mag_f64_every8"] = m64[::4, ::8].copy()        # float64 definition
        ext[f"{tag}_db_every8"] = sdb[::4, ::8].copy()
        ext[f"{tag}_proj_every8"] = proj[::8, ::8].copy()
        ext[f"{tag}_x_every97"] = x[::97].copy()
    np.savez_compressed(os.path.join(HERE, "extract.npz"), **ext)

    # ---- search: ragged database, planted / tied / over-long queries ----
    lens = [300, 41, 1, 120, 300, 77, 5, 300]
    db = gen.golden_u64(sum(lens), 9)
    db_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    db[db_off[4]:db_off[5]] = db[db_off[0]:db_off[1]]              # clip 4 duplicates clip 0 (ties)
    qs = [db[20:60].copy(), db[db_off[3] + 10: db_off[3] + 110].copy(), gen.golden_u64(50, 1234),
          gen.golden_u64(200, 99), db[db_off[7] + 250: db_off[7] + 300].copy()]
    qs[1] ^= np.uint64(0x0000010000000001)                         # two bit flips per hashprint
    q = np.concatenate(qs)
    q_off = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int64)
    top = oracle.search_topk(db, db_off, q, q_off, 5)
    want = brute_topk(db, db_off, q, q_off, 5)
    for qi, rows in enumerate(want):
        for t, (d, c, off) in enumerate(rows):
            assert (top[qi, t]["dist"], top[qi, t]["clip"], top[qi, t]["offset"]) == (d, c, off), (qi, t)
    np.savez_compressed(os.path.join(HERE, "search.npz"), db=db, db_off=db_off, q=q, q_off=q_off,
                        top5=top)
    for f in ("extract.npz", "search.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
