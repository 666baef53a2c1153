#!/bin/bash
# the file-level FFI section of bench.py alone and behind the host-buffer (pcie) section: distinct-length files per second
cd "$(dirname "$0")/.."
COMMON="--steps 3 --warmup 1 --no-search --no-stream --no-learn --no-cpu-baseline --no-f32-chain --no-any-length --no-parity"
for rep in 1 2; do
for extra in "--no-pcie" ""; do
  HPFW_PLAN_TIMING=1 timeout -k 10 300 python bench.py $COMMON $extra 2> >(grep "plan timing" >&2) | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln)['ffi']; print('pcie section: ${extra:-yes}', {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if 'per_s' in kk}) for k, v in d.items() if k in ('equal_lengths','distinct_lengths')})
"
done
done
