#!/bin/bash
# kernel times of the extraction step, twice (one box): tools/quick_bench.sh [reps]
cd "$(dirname "$0")/.."
for rep in $(seq ${1:-2}); do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-search --no-pcie --no-any-length --no-learn --no-f32-chain --no-ffi --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['ms_per_step'], d['kernel_ms_one_pass'], d.get('parity'))
"
done
