#!/bin/bash
# search-only timing: bench.py's search section
cd "$(dirname "$0")/.."
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --clips 64 "$@" > gpurun_out/qs.log 2> gpurun_out/qs.err
python3 - <<'PY'
import json
for ln in open('gpurun_out/qs.log'):
    if ln.startswith('{'):
        d = json.loads(ln)['search']
        print(d['ms_per_search'], f"{d['value']:.4g}", d['planted_queries_found'], d.get('scan_kernel'))
PY
tail -2 gpurun_out/qs.err
