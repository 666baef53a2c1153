#!/bin/bash
# separate --pmc passes over the search workload; output under gpurun_out/pmcs/<group>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmcs
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 tools/search_pass.py 4000 1000 1 > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/g$i.log; }
  echo "group $i done"
done
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmcs/g*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'hamming' not in k and 'topk' not in k and 'expand' not in k: continue
        k = re.sub(r'^void ', '', k).split('(')[0].replace('hpfw::','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k, {c: round(sum(v)/len(v)) for c, v in agg[k].items()})
PY
