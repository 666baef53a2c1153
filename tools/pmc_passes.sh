#!/bin/bash
# separate --pmc passes (kernel-trace only), one counter group each; output under gpurun_out/pmc/<group>
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p $OUT
CLIPS=${1:-500}
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 tools/extract_pass.py $CLIPS 2 > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/g$i.log; }
  echo "group $i done"
done
