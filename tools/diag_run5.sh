#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag5
mkdir -p $O
rm -f $O/share.log
for cfg in "1 0" "1 1" "2 1" "3 1" "4 1"; do
  timeout -k 10 120 tools/simd_share_probe.bin $cfg 20 >> $O/share.log 2>&1
done
cat $O/share.log
timeout -k 10 300 python tools/rows_error_shape.py > $O/rows_shape.log 2> $O/rows_shape.err
tail -3 $O/rows_shape.err; tail -40 $O/rows_shape.log
