"""Integer-only deterministic inputs for the golden fixtures (no libm, no RNG library: the same
arrays on any machine forever)."""
import numpy as np


def golden_pcm(n, seed):
    """int16 'audio': four triangle waves with integer phase steps inside the constant-Q range plus
    hashed noise."""
    i = np.arange(n, dtype=np.int64)
    x = np.zeros(n, np.int64)
    for k, (step, amp) in enumerate([(331 + 17 * seed, 5000), (977 + 5 * seed, 4000), (2203 + seed, 3000),
                                     (4801 + 3 * seed, 2000)]):
        ph = (i * step + 12345 * k) % 65536
        x += (np.abs(ph - 32768) - 16384) * amp // 16384
    # amplitude modulation in 0.2 s blocks so that the spectrogram changes over time
    blk = (i // 8820 + seed) % 7
    x = x * (3 + blk) // 9
    noise = (((i * 2654435761 + seed * 40503) >> 7) & 0x3FF) - 512
    return np.clip(x + noise, -32768, 32767).astype(np.int16)


def golden_u64(n, seed):
    """splitmix64 of the index: exact integer arithmetic modulo 2^64"""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def golden_filters():
    """64 x 2420 filter fixture from the integer hash, rows scaled to about unit norm; flat
    column-major (r, k) at r + 64 k.  Every value is a multiple of 2^-24 (exact in float32)."""
    h = golden_u64(64 * 2420, 77)
    v = ((h >> np.uint64(40)).astype(np.int64) - (1 << 23)).astype(np.float64) / (1 << 23)   # [-1, 1)
    return (v * (1.0 / 32.0)).astype(np.float32)
