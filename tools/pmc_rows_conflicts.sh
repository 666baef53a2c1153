#!/bin/bash
# LDS bank conflicts of the row kernel alone: one --pmc pass over a light bench run (stages launched whole)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HPFW_FWD_CHUNK=0
rm -rf gpurun_out/pmc_rows && mkdir -p gpurun_out/pmc_rows
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_rows -o p --output-format csv -- python3 bench.py --no-parity --no-cpu-baseline --no-pcie --no-any-length --no-learn --no-search --no-stream --no-ffi --no-f32-chain --steps 2 --warmup 1 --batch 1000 > gpurun_out/pmc_rows/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_rows/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "fwd_rows2" in k or "cq_kernel<12288" in k or "cq_kernel<6144" in k:
        acc[(k.split("(")[0][-60:], int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, g), d in sorted(acc.items(), key=lambda kv: -kv[0][1])[:6]:
    print(k, g, "conflict fraction %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)), "lds busy %.3f" % (d["SQ_LDS_IDX_ACTIVE"] / max(d["SQ_BUSY_CYCLES"], 1) / 8))
PY
rm -rf gpurun_out/pmc_rows
