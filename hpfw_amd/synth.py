"""Deterministic synthetic inputs (there is no network for datasets): 44.1 kHz PCM16 "songs" made
of short notes with a few partials inside the constant-Q range plus noise, query slices cut from
them, and a filter fixture.  Used by tests/ and bench.py; `data` is always "synthetic"."""
import struct

import numpy as np

SR = 44100
SEED = 0x68706677  # "hpfw"
FMIN, FMAX = 130.81, 4186.01


def gen_clip(clip_id, seconds=30.0, seed=SEED):
    """int16 mono clip: 0.25 s notes, 6 partials each (log-uniform in the CQ range), 10 ms fades,
    white noise at -30 dBFS."""
    rng = np.random.default_rng([seed, int(clip_id)])
    n = int(round(seconds * SR))
    note = SR // 4
    x = np.zeros(n, np.float64)
    t = np.arange(note) / SR
    fade = np.minimum(1.0, np.minimum(np.arange(note), note - 1 - np.arange(note)) / (0.010 * SR))
    for s in range(0, n, note):
        m = min(note, n - s)
        f = FMIN * (FMAX / FMIN) ** rng.random(6)
        a = rng.uniform(0.05, 0.2, 6)
        ph = rng.uniform(0, 2 * np.pi, 6)
        seg = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :m] + ph[:, None])).sum(0)
        x[s:s + m] += seg * fade[:m]
    x += 10 ** (-30 / 20) * rng.standard_normal(n)
    return np.clip(np.round(x * 32767 / max(1.0, np.abs(x).max())), -32768, 32767).astype(np.int16)


def gen_query(clips, q, seconds=5.0, seed=SEED, snr_db=10.0):
    """Query q: a slice of clip q mod n_clips starting at 44100 * (3 + (7 q) mod 20) samples
    (clamped to the clip), gain 0.5, white noise at snr_db.  Returns (pcm, clip index, start)."""
    n_clips = len(clips)
    ci = q % n_clips
    src = clips[ci].astype(np.float64)
    n = int(round(seconds * SR))
    start = SR * (3 + (7 * q) % 20)
    start = max(0, min(start, src.size - n))
    seg = 0.5 * src[start:start + n]
    rng = np.random.default_rng([seed ^ 0x9E3779B9, int(q)])
    p = float(np.mean(seg ** 2)) + 1e-12
    seg = seg + np.sqrt(p / 10 ** (snr_db / 10)) * rng.standard_normal(n)
    return np.clip(np.round(seg), -32768, 32767).astype(np.int16), ci, start


def make_filters(seed=SEED):
    """64 orthonormal rows of length 2420 (stand-in for the learned eigenvectors, which are an input
    fixture everywhere).  Returned flat in the reference's column-major layout: (r, k) at r + 64 k."""
    rng = np.random.default_rng([seed, 64, 2420])
    a = rng.standard_normal((2420, 64))
    qm, _ = np.linalg.qr(a)            # [2420][64], orthonormal columns
    return np.ascontiguousarray(qm, np.float32).ravel()   # row k holds the 64 filters: k*64 + r


def filters_rows(filters_colmajor):
    """[64][2420] view (row r, column k) of the flat column-major fixture"""
    return np.asarray(filters_colmajor, np.float32).reshape(2420, 64).T


def random_hashprints(n_clips, n_hp, seed=SEED):
    rng = np.random.default_rng([seed, 3, int(n_clips), int(n_hp)])
    return rng.integers(0, 2 ** 64, size=(n_clips, n_hp), dtype=np.uint64)


def planted_queries(db, n_q, k, flip_bits=6, seed=SEED):
    """Queries cut from db[q mod n_clips] at a pseudo-random offset with `flip_bits` random bit
    flips per hashprint.  Returns (queries [n_q][k], clip ids, offsets)."""
    rng = np.random.default_rng([seed, 4, int(n_q), int(k)])
    n_clips, n_hp = db.shape
    qs = np.zeros((n_q, k), np.uint64)
    cids = np.zeros(n_q, np.int64)
    offs = np.zeros(n_q, np.int64)
    for q in range(n_q):
        c = q % n_clips
        o = int(rng.integers(0, n_hp - k + 1))
        seg = db[c, o:o + k].copy()
        for _ in range(flip_bits):
            seg ^= np.uint64(1) << rng.integers(0, 64, size=k, dtype=np.uint64)
        qs[q], cids[q], offs[q] = seg, c, o
    return qs, cids, offs


def write_wav(path, pcm, channels=1):
    pcm = np.ascontiguousarray(pcm, np.int16)
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, SR, SR * 2 * channels, 2 * channels, 16))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)
