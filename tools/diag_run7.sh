#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag7
mkdir -p $O
echo "== interference, no packed arithmetic (plain complex helpers, -fno-slp-vectorize)"
HPFW_GPU_LIB=$PWD/hpfw_amd/lib_nopk/libhpfw_gpu.so timeout -k 10 400 python tools/interfere.py 12 > $O/interfere_nopk.jsonl 2> $O/interfere_nopk.err
tail -2 $O/interfere_nopk.err; cat $O/interfere_nopk.jsonl
