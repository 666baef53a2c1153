#!/bin/bash
# the any-length (chirp-z forward transform) section of bench.py for several library builds: tools/ab_any_length.sh lib lib_old
cd "$(dirname "$0")/.."
ARGS="--steps 3 --warmup 1 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain"
for rep in 1 2; do
for v in "$@"; do
  HPFW_GPU_LIB=$PWD/hpfw_amd/$v/libhpfw_gpu.so timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln)['any_length']; print('$v', {k: v for k, v in d.items() if k != 'note'})
" | cut -c1-400
done
done
