"""calc_hashprints over N equal-length files on tmpfs, repeated (run on the GPU box): python tools/ffi_equal.py [files] [reps]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hpfw_amd
from hpfw_amd import synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1323000
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
cache = os.path.join(d, "cache") + "/"; os.makedirs(cache)
filt = synth.make_filters()
open(os.path.join(cache, "filters.cereal"), "wb").write(np.array([64, 2420], np.int32).tobytes() + np.ascontiguousarray(filt, np.float32).tobytes())
pc = hpfw_amd.ParallelCollector(); pc.load(cache)
src = [synth.gen_clip(i + 1, 30.0)[:n] for i in range(8)]
paths = []
for i in range(nf):
    p = os.path.join(d, f"t{i:05d}.wav"); synth.write_wav(p, src[i % 8]); paths.append(p)
pc.calc_hashprints(paths[:256])
rates = []
for r in range(reps):
    t0 = time.perf_counter(); got = pc.calc_hashprints(paths); dt = time.perf_counter() - t0
    rates.append(round(len(got) / dt))
print(os.environ.get("HPFW_FFI_NO_OVERLAP", "overlap"), rates)
shutil.rmtree(d)
