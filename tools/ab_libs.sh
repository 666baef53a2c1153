#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh lib lib_w4 ...   (each: the quick bench, two runs)
cd "$(dirname "$0")/.."
ARGS="--steps 10 --warmup 3 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length"
for rep in 1 2; do
for v in "$@"; do
  HPFW_GPU_LIB=$PWD/hpfw_amd/$v/libhpfw_gpu.so timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$v', d['ms_per_step'], d['kernel_ms_one_pass'], d.get('parity',{}).get('bit_identical'))
"
done
done
