#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
for d in 0 4 8 16 1000; do
  HPFW_CQ_ROWS_MIN=$d timeout -k 10 300 python bench.py --no-cpu-baseline --no-search --no-parity --no-pcie --no-any-length --no-learn --no-f32-chain --no-ffi --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('rows_min=$d', d['ms_per_step'], d['kernel_ms_one_pass'])
"
done
done
