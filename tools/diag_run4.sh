#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag4
mkdir -p $O
rm -f $O/probe2.log
P=tools/corrupt_probe.bin
for cfg in "50400 512 30 1 1" "50400 512 30 0 1" "26112 256 30 1 1" "50400 512 30 1 2" "26112 256 30 1 2" "50400 512 30 1 3" "26112 256 30 1 3"; do
  echo "== $cfg" >> $O/probe2.log
  timeout -k 10 120 $P $cfg >> $O/probe2.log 2>&1
done
cat $O/probe2.log
