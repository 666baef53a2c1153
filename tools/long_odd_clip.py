"""Long clips whose length has a prime factor above 7 (the chirp-z forward transform with a large n1): the dB
spectrogram against the float64 definition (numpy FFT of the exact length; the C oracle's dense column DFT would take
minutes here) and the throughput.  python3 tools/long_odd_clip.py [seconds] [clips]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402
from oracle import nsgt_f64  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(round(seconds * 44100)) + 1
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
t0 = time.perf_counter()
geo = g.geometry(n)
print(f"N = {n}: n1 {geo.n1} n2 {geo.n2} M {geo.m} C {geo.c}; plan + tables {time.perf_counter() - t0:.2f} s", flush=True)
clip = synth.gen_clip(3, seconds + 0.01)[:n]
nk = geo.kmax - geo.kmin
pcm = torch.from_numpy(np.stack([clip] * n_clips)).cuda()
d_x = torch.empty((1, nk, 2), dtype=torch.float32, device="cuda")
d_mag = torch.empty((1, 121, geo.c), dtype=torch.float32, device="cuda")
g.stage_spectrum_dev(pcm.data_ptr(), n, 1, d_x.data_ptr())
g.stage_cqmag_dev(d_x.data_ptr(), n, 1, d_mag.data_ptr())
torch.cuda.synchronize()
x = d_x.cpu().numpy()[0]
ref = np.fft.fft(clip.astype(np.float64) / 32768.0)[geo.kmin:geo.kmax]
err = np.abs((x[:, 0] + 1j * x[:, 1]) - ref)
print(f"forward bins vs numpy float64: max {err.max() / np.abs(ref).max():.2e} of the largest bin, "
      f"rms {np.sqrt((err ** 2).mean()) / np.sqrt((np.abs(ref) ** 2).mean()):.2e}", flush=True)
m64 = nsgt_f64.cq_magnitudes(clip)
rel = (np.abs(d_mag.cpu().numpy()[0] - m64).max(axis=1) / m64.max(axis=1)).max()
print(f"|CQ| vs float64 definition: {rel:.2e} of each band's maximum (bar 1e-4)", flush=True)
hp = torch.zeros((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
g.set_kernel_timing(-1)
t0 = time.perf_counter()
g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n_clips} clips: {dt * 1e3:.1f} ms = {n_clips * seconds / dt:.0f} x real time; kernels "
      f"{ {k: round(v[0], 2) for k, v in g.kernel_timing().items() if v[1]} }")
sys.exit(0 if rel < 1e-4 else 1)
