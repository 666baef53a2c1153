#!/bin/bash
# per-kernel average durations of an arbitrary python command: tools/kstats_cmd.sh tools/time_learn.py 256
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o p --output-format csv -- python3 "$@" > gpurun_out/ks/run.log 2>&1
python3 - <<'PY'
import csv, re
for r in csv.DictReader(open('gpurun_out/ks/p_kernel_stats.csv')):
    if 'hpfw::' in r['Name']:
        n = re.sub(r'^void ', '', r['Name']).split('(')[0].replace('hpfw::', '')[:60]
        print(f"{n:62s} calls {r['Calls']:>3s}  avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
rm -f gpurun_out/ks/p_kernel_trace.csv
