#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag8
mkdir -p $O
ARGS="--steps 10 --warmup 3 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length"
for v in lib lib_nopk; do
  HPFW_GPU_LIB=$PWD/hpfw_amd/$v/libhpfw_gpu.so timeout -k 10 300 python bench.py $ARGS > $O/bench_$v.json 2> $O/bench_$v.err
  python - "$O/bench_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d.get("parity"), {k: d[k] for k in d if "kernel_ms" in k or "stage" in k})
PY
done
