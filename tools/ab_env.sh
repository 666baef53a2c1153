#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR VALUE [reps]   (unset vs VAR=VALUE, alternating)
cd "$(dirname "$0")/.."
VAR=$1; VAL=${2:-1}; REPS=${3:-2}
for rep in $(seq $REPS); do
for v in "" "$VAL"; do
  env ${v:+$VAR=$v} timeout -k 10 300 python bench.py --no-cpu-baseline --no-search --no-pcie --no-any-length --no-learn --no-f32-chain --no-ffi --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$VAR=${v:-unset}', d['ms_per_step'], d['kernel_ms_one_pass'], d.get('parity',{}).get('bit_identical'))
"
done
done
