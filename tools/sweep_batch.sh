#!/bin/bash
# clips per front-end pass: tools/sweep_batch.sh 250 125 500 1000   (bench.py --batch; the default, 256, gives four passes of 250)
cd "$(dirname "$0")/.."
ARGS="--steps 10 --warmup 3 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length --no-parity"
for rep in 1 2; do
for b in "$@"; do
  timeout -k 10 300 python bench.py $ARGS --batch $b 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernel_ms_one_pass']
        print('batch $b', 'step', d['ms_per_step'], 'fwd_span', k['fwd_span'], 'cq', k['cq_chirpz'], 'project', k['project_mfma'])
"
done
done
