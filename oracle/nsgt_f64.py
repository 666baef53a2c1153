"""Float64 numpy statement of the mathematics behind the hot path -- the independent pin of the
C oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED against the reference, see hpfw_oracle.h).

It follows the published algorithm directly, with none of the C oracle's decomposition:
one full-length FFT, one length-M inverse FFT per band, every third sample.

  essentia NSGConstantQ as configured at /root/reference/include/hpfw/spectrum/cqt.h:54-61
  (Holighaus/Doerfler/Velasco/Grill NSGT; essentia nsgconstantq.cpp designWindow/compute):
    f_j = fmin 2^(j/24), bw_j = (2^(1/24) - 2^(-1/24)) f_j, fftres = sr / N
    posit_j = floor(f_j / fftres), Lg_j = max(round(bw_j / fftres), minimumWindow = 96)
    window = hann(Lg_j) = 0.5 - 0.5 cos(2 pi i / (Lg_j - 1)), rasterize "full": M = max_j Lg_j
    c_j = IFFT_M( circular placement of X[posit_j - floor(Lg_j/2) + i] * hann[i] )
  hpfw keeps |c_j[3 c]| (cqt.h:73-81) and converts to dB (convert.h:7-25).
"""
import numpy as np

SR = 44100.0
FMIN, FMAX = 130.81, 4186.01
BPO = 24
MIN_WINDOW = 96
BINS = 121


# essentia conventions that cannot be checked offline (same bits as HPFW_CONV_* of include/hpfw_gpu.h)
CONV_HANN_PERIODIC, CONV_LG_HALF_EVEN, CONV_FLOAT_GEOMETRY, CONV_NO_IFFT_SCALE = 1, 2, 4, 8


def bands(n, conventions=0):
    nb = int(np.floor(BPO * np.log2(FMAX / FMIN))) + 1
    assert nb == BINS
    if conventions & CONV_FLOAT_GEOMETRY:                     # essentia's Real is float
        f32 = np.float32
        fftres = f32(SR) / f32(n)
        q = f32(2.0) ** (f32(1.0) / f32(BPO)) - f32(2.0) ** (f32(-1.0) / f32(BPO))
        f = f32(FMIN) * f32(2.0) ** (np.arange(nb, dtype=np.float32) / f32(BPO))
        posit = np.floor(f / fftres).astype(np.int64)
        bw = (q * f / fftres).astype(np.float64)
    else:
        fftres = SR / n
        q = 2.0 ** (1.0 / BPO) - 2.0 ** (-1.0 / BPO)
        f = FMIN * 2.0 ** (np.arange(nb) / BPO)
        posit = np.floor(f / fftres).astype(np.int64)
        bw = q * f / fftres
    rounded = np.rint(bw) if conventions & CONV_LG_HALF_EVEN else np.floor(bw + 0.5)
    lg = np.maximum(rounded.astype(np.int64), MIN_WINDOW)
    return posit, lg


def cq_magnitudes(pcm, conventions=0):
    """|c_j[3c]| in float64, bin-major [121][ceil(M/3)], for int16 PCM."""
    x = np.asarray(pcm, np.float64) / 32768.0
    n = x.size
    spec = np.fft.fft(x)
    posit, lg = bands(n, conventions)
    m = int(lg.max())
    cols = (m + 2) // 3
    out = np.zeros((BINS, cols))
    for j in range(BINS):
        L = int(lg[j])
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(L) / (L if conventions & CONV_HANN_PERIODIC else L - 1))
        idx = (posit[j] - L // 2 + np.arange(L)) % n
        prod = spec[idx] * win
        buf = np.zeros(m, np.complex128)
        # essentia places the upper half of the slice at the start and the lower half at the end
        # of the length-M buffer (band centre at index 0); only the phase depends on this.
        half = L // 2
        buf[: L - half] = prod[half:]
        buf[m - half:] = prod[:half]
        cj = np.fft.ifft(buf)  # 1/M normalised
        if conventions & CONV_NO_IFFT_SCALE:
            cj = cj * m
        out[j] = np.abs(cj[::3])[:cols]
    return out


def amplitude_to_db(mag):
    """convert.h:18-25 -> 7-16 in float64."""
    p = np.asarray(mag, np.float64) ** 2
    mx = max(1e-10, float(p.max()))
    log_spec = 10.0 * np.log10(np.where(p < 1e-10, 1e-10, p)) - 10.0 * np.log10(mx)
    mx2 = log_spec.max()
    return np.where(log_spec < mx2 - 80.0, mx2 - 80.0, log_spec)


def project(filters_rk, s_db):
    """hashprint_handle.h:79-93 + parallel_collector.h:57: filters [64][2420] (row r, col k),
    s_db [121][C] -> [64][C-19] in float64."""
    s = np.asarray(s_db, np.float64)
    c = s.shape[1]
    nf = c - 19
    frames = np.zeros((BINS * 20, nf))
    for b in range(BINS):
        for t in range(20):
            frames[b * 20 + t] = s[b, t:t + nf]
    return np.asarray(filters_rk, np.float64) @ frames


def pack(proj):
    """hashprint_handle.h:115-142: delta over 80 frames, >= 0, MSB-first packing."""
    d = proj[:, :-80] - proj[:, 80:]
    bits = (d >= 0)
    w = (np.uint64(1) << np.arange(63, -1, -1, dtype=np.uint64))
    return (bits.astype(np.uint64) * w[:, None]).sum(axis=0, dtype=np.uint64)


def frames(s_db):
    """hashprint_handle.h:79-93: frames[b*20 + t, n] = S[b, n + t], float64 [2420][C-19]"""
    s = np.asarray(s_db, np.float64)
    nf = s.shape[1] - 19
    out = np.zeros((BINS * 20, nf))
    for b in range(BINS):
        for t in range(20):
            out[b * 20 + t] = s[b, t:t + nf]
    return out


def covariance(s_db):
    """hashprint_handle.h:96-102 on frames^T: centre every component on its mean over the frames,
    centred^T centred / (n_frames - 1); float64 [2420][2420]"""
    x = frames(s_db)
    xc = x - x.mean(axis=1, keepdims=True)
    return xc @ xc.T / (x.shape[1] - 1)


# ---- f3: the Mel front-end (mel.h:34-104) in float64, straight from the restated essentia algorithms ----
MEL_BANDS, MEL_FRAME, MEL_HOP = 33, 4410, 441


def mel_filterbank():
    """essentia MelBands(inputSize 2206, numberBands 33; 0..22050 Hz, htkMel, weighting "warping",
    normalize "unit_sum") -> [33][2206]"""
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)   # noqa: E731
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)  # noqa: E731
    nb = MEL_FRAME // 2 + 1
    fb = mel2hz(np.linspace(hz2mel(0.0), hz2mel(22050.0), MEL_BANDS + 2))
    fscale = 22050.0 / (nb - 1)
    bf = np.arange(nb) * fscale
    out = np.zeros((MEL_BANDS, nb))
    for i in range(MEL_BANDS):
        jb, je = int(fb[i] / fscale + 0.5), min(int(fb[i + 2] / fscale + 0.5), nb - 1)
        j = np.arange(jb, je + 1)
        up = (bf[j] >= fb[i]) & (bf[j] < fb[i + 1])
        dn = (bf[j] >= fb[i + 1]) & (bf[j] < fb[i + 2])
        c = np.where(up, (hz2mel(bf[j]) - hz2mel(fb[i])) / (hz2mel(fb[i + 1]) - hz2mel(fb[i])), 0.0)
        c = np.where(dn, (hz2mel(fb[i + 2]) - hz2mel(bf[j])) / (hz2mel(fb[i + 2]) - hz2mel(fb[i + 1])), c)
        out[i, j] = c / c.sum()
    return out


def mel_power(pcm):
    """band powers of every frame [33][n_frames] and the keep mask (frames that are not silent)"""
    x = np.asarray(pcm, np.float64) / 32768.0
    n = x.size
    nfr = (n + MEL_FRAME // 2 + MEL_HOP - 1) // MEL_HOP
    pad = np.concatenate([np.zeros(MEL_FRAME // 2), x, np.zeros(MEL_FRAME + MEL_HOP)])
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(MEL_FRAME) / (MEL_FRAME - 1))
    win *= 2.0 / win.sum()
    fbk = mel_filterbank()
    power = np.zeros((MEL_BANDS, nfr))
    keep = np.zeros(nfr, bool)
    for f in range(nfr):
        fr = pad[f * MEL_HOP: f * MEL_HOP + MEL_FRAME]
        keep[f] = (fr * fr).sum() / MEL_FRAME >= 1e-10
        spec = np.abs(np.fft.rfft(np.roll(fr * win, -(MEL_FRAME // 2))))   # zero-phase windowing, magnitude
        power[:, f] = fbk @ (spec * spec)
    return power, keep


def power_to_db(p):
    """convert.h:7-16 in float64"""
    p = np.asarray(p, np.float64)
    mx = max(1e-10, float(p.max())) if p.size else 1e-10
    log_spec = 10.0 * np.log10(np.maximum(p, 1e-10)) - 10.0 * np.log10(mx)
    return np.maximum(log_spec, log_spec.max() - 80.0) if p.size else log_spec
