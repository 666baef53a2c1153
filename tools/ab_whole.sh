#!/bin/bash
# stage times with the forward transform launched whole (HPFW_FWD_CHUNK=0: no overlap between its kernels): tools/ab_whole.sh lib ...
cd "$(dirname "$0")/.."
ARGS="--steps 6 --warmup 2 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length --no-parity"
for rep in 1 2; do
for v in "$@"; do
  HPFW_FWD_CHUNK=0 HPFW_GPU_LIB=$PWD/hpfw_amd/$v/libhpfw_gpu.so timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$v whole', d['ms_per_step'], d['kernel_ms_one_pass'])
"
done
done
