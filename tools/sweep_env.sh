#!/bin/bash
# the quick bench under several values of one environment switch, alternating, with each run's stage times normalised by
# the hashprint kernel's time of the same run (the boxes drift by a few per cent from run to run):
#   tools/sweep_env.sh HPFW_FWD_CHUNK 16 8 32 0
cd "$(dirname "$0")/.."
VAR=$1; shift
ARGS="--steps 10 --warmup 3 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length --no-parity"
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernel_ms_one_pass']; p=k['project_mfma']
        print('$VAR=$v', 'step', d['ms_per_step'], 'step/project %.3f' % (d['ms_per_step']/p), 'fwd_span', k['fwd_span'], 'cq', k['cq_chirpz'], 'project', p)
"
done
done
