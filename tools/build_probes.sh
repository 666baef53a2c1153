#!/bin/bash
# the stand-alone probes of tools/ (not part of the library), built for gfx950; the .bin files travel to the GPU box
cd "$(dirname "$0")"
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize"
L="-L../hpfw_amd/lib -lhpfw_gpu -Wl,-rpath,\$ORIGIN/../hpfw_amd/lib"
$H -o pk_mfma_repro.bin pk_mfma_repro.hip $L       # packed FP32 beside int8 matrix kernels: self-checking reproducer
$H -o corrupt_probe.bin corrupt_probe.hip $L       # LDS / registers / loads / exchanges beside hashprint_q_kernel: all clean
$H -o simd_share_probe.bin simd_share_probe.hip    # waves of ONE kernel sharing a SIMD with int8 matrix waves: clean
$H -o cwsr_probe.bin cwsr_probe.hip                # state kept across context save/restore (queues of other processes): clean
