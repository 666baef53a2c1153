#!/bin/bash
# SQ counters of one kernel (name substring $1) for an arbitrary python command, in separate --pmc passes:
#   tools/pmc_cmd.sh hamming_shift tools/search_pass.py 125000 1 1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
KERN=$1; shift
OUT=gpurun_out/pmcc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
KERN=$KERN python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmcc/g*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if os.environ['KERN'] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print(f"{c:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
find $OUT -name "*_kernel_trace.csv" -delete
