#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag6
mkdir -p $O
HPFW_GPU_LIB=$PWD/hpfw_amd/lib_snap/libhpfw_gpu.so timeout -k 10 400 python tools/rows_snapshots.py > $O/snap.log 2> $O/snap.err
tail -5 $O/snap.err; head -150 $O/snap.log
