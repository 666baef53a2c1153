#!/bin/bash
# the chirp-z forward path's kernels launched whole (no chunks, nothing beside them): rocprofv3's average durations
#   tools/bz_whole_stats.sh [n_samples] [clips] [library directory under hpfw_amd/]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/bzw
[ -n "$3" ] && export HPFW_GPU_LIB=$PWD/hpfw_amd/$3/libhpfw_gpu.so && echo "== $3"
HPFW_BZ_CHUNK=0 HPFW_CQ_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/bzw -o t --output-format csv -- python3 tools/chirpz_profile.py ${1:-1323001} ${2:-256} 5 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/bzw/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1000:9.1f} us  {r["Percentage"]} %')
PY
rm -rf gpurun_out/bzw
