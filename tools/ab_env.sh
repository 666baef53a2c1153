#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR [reps]   (unset vs VAR=1, alternating)
cd "$(dirname "$0")/.."
VAR=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
for v in "" 1; do
  env ${v:+$VAR=$v} timeout -k 10 300 python bench.py --no-cpu-baseline --no-search --no-parity --no-pcie --no-any-length --no-learn --no-f32-chain --no-ffi --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$VAR=${v:-unset}', d['ms_per_step'], d['kernel_ms_one_pass'])
"
done
done
