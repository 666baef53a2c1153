#!/bin/bash
# extraction-only timing: bench.py's JSON reduced to the per-kernel line
cd "$(dirname "$0")/.."
timeout -k 10 600 python bench.py --no-cpu-baseline --no-search "$@" > gpurun_out/qb.log 2> gpurun_out/qb.err
python3 - <<'PY'
import json
for ln in open('gpurun_out/qb.log'):
    if ln.startswith('{'):
        d = json.loads(ln)
        print(d['clips_per_s'], d['ms_per_step'], d['kernel_ms_one_pass'], d['roofline']['frac'], d.get('parity'))
PY
tail -3 gpurun_out/qb.err
