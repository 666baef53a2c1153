"""power_clock.py -- socket power and shader clock while each stage of the extraction runs in a loop (librocm_smi64, read
only).  Why: the stages do not gain from running side by side, and fewer matrix instructions did not make hashprint_q_kernel
faster (MEASURED_NOT_KEPT.md) -- is it a power budget the kernels share?   python tools/power_clock.py [seconds per stage]"""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpfw_amd  # noqa: E402
from hpfw_amd import synth  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
smi = ctypes.CDLL("librocm_smi64.so")


class Freqs(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


assert smi.rsmi_init(ctypes.c_uint64(0)) == 0


def sample():
    p = ctypes.c_uint64(0)
    ok_p = smi.rsmi_dev_current_socket_power_get(0, ctypes.byref(p)) == 0 or smi.rsmi_dev_power_ave_get(0, 0, ctypes.byref(p)) == 0
    f = Freqs()
    ok_f = smi.rsmi_dev_gpu_clk_freq_get(0, 0, ctypes.byref(f)) == 0
    mhz = f.frequency[f.current] / 1e6 if ok_f and f.current < 33 else float("nan")
    return (p.value / 1e6 if ok_p else float("nan")), mhz


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.run_flag, self.rows = True, []

    def run(self):
        while self.run_flag:
            self.rows.append(sample())
            time.sleep(0.004)


n_clips, n = 250, 1323000
g = hpfw_amd.Gpu(0)
g.set_filters(synth.make_filters())
geo = g.geometry(n)
gen = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n_clips, n, device="cuda", generator=gen) * 3000).to(torch.int16)
nk = geo.kmax - geo.kmin
x = torch.zeros((n_clips, nk, 2), dtype=torch.float32, device="cuda")
mag = torch.zeros((n_clips, 121, geo.c), dtype=torch.float32, device="cuda")
db = torch.zeros_like(mag)
hp = torch.zeros((n_clips, geo.n_hp), dtype=torch.int64, device="cuda")
g.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, x.data_ptr())
g.stage_cqmag_dev(x.data_ptr(), n, n_clips, mag.data_ptr())
g.stage_db_dev(mag.data_ptr(), n_clips, geo.c, db.data_ptr())
big_a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
big_b = torch.empty_like(big_a)
torch.cuda.synchronize()
STAGES = [
    ("idle", lambda: time.sleep(0.01)),
    ("forward transform (column + row stage)", lambda: g.stage_spectrum_dev(pcm.data_ptr(), n, n_clips, x.data_ptr())),
    ("chirp-z stage", lambda: g.stage_cqmag_dev(x.data_ptr(), n, n_clips, mag.data_ptr())),
    ("hashprint_q_kernel", lambda: g.hashprints_from_db_dev(db.data_ptr(), n_clips, geo.c, hp.data_ptr())),
    ("whole extraction", lambda: g.extract_dev(pcm.data_ptr(), n, n_clips, hp.data_ptr())),
    ("HBM copy 1 GB", lambda: big_b.copy_(big_a)),
]
for name, fn in STAGES:
    fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    calls = 0
    while time.perf_counter() - t0 < SECS:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        calls += 4
    dt = time.perf_counter() - t0
    s.run_flag = False
    s.join()
    a = np.array(s.rows[len(s.rows) // 4:])            # the first quarter: the sensors' averaging window filling
    print(f"{name:42s} {1e3 * dt / calls:8.3f} ms per call   power {np.nanmean(a[:, 0]):6.0f} W (max {np.nanmax(a[:, 0]):4.0f})   "
          f"shader clock {np.nanmean(a[:, 1]):6.0f} MHz (min {np.nanmin(a[:, 1]):5.0f})   {len(a)} samples", flush=True)
