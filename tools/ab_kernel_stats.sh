#!/bin/bash
# per-kernel average durations of library builds on one box:  tools/ab_kernel_stats.sh <kernel name pattern> lib lib_old ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PAT="$1"; shift
ARGS="--steps 10 --warmup 3 --no-search --no-stream --no-learn --no-cpu-baseline --no-pcie --no-ffi --no-f32-chain --no-any-length"
for rep in 1 2; do
for v in "$@"; do
  d=gpurun_out/abk_${v}_$rep
  rm -rf $d
  HPFW_GPU_LIB=$PWD/hpfw_amd/$v/libhpfw_gpu.so timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o t --output-format csv -- python3 bench.py $ARGS > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v run $rep"
  python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1000:8.1f} us')
PY
done
done
