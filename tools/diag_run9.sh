#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag9
mkdir -p $O
rm -f $O/repro.jsonl
for nb in 0 1 2 3 4 5 6; do
  timeout -k 10 120 tools/pk_mfma_repro.bin 20 $nb 2>&1 | head -c 1500 | tr '\n' ' ' >> $O/repro.jsonl; echo >> $O/repro.jsonl
done
cut -c1-260 $O/repro.jsonl
