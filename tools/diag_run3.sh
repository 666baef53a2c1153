#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag3
mkdir -p $O
echo "== interference, product build"
timeout -k 10 400 python tools/interfere.py 12 > $O/interfere.jsonl 2> $O/interfere.err
tail -2 $O/interfere.err; cat $O/interfere.jsonl
echo "== interference, plain-complex build"
HPFW_GPU_LIB=$PWD/hpfw_amd/lib_plain/libhpfw_gpu.so timeout -k 10 400 python tools/interfere.py 12 > $O/interfere_plain.jsonl 2> $O/interfere_plain.err
tail -2 $O/interfere_plain.err; cat $O/interfere_plain.jsonl
