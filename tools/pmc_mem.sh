#!/bin/bash
# memory-side counters of one kernel (name substring $1) for an arbitrary python command, in separate --pmc passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
KERN=$1; shift
OUT=gpurun_out/pmcm
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 "$@" > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
KERN=$KERN python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmcm/g*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if os.environ['KERN'] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print(f"{c:32s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
find $OUT -name "*_kernel_trace.csv" -delete
