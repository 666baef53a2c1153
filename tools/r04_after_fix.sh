#!/bin/bash
# round 4, after the fix: the reproducer's neighbours, the interference matrix on the product build, the full GPU suite
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag10
mkdir -p $O
rm -f $O/repro.jsonl
for nb in 0 1 2 3 4 5 6; do
  timeout -k 10 120 tools/pk_mfma_repro.bin 20 $nb 2>&1 | head -c 1500 | tr '\n' ' ' >> $O/repro.jsonl; echo >> $O/repro.jsonl
done
cut -c1-230 $O/repro.jsonl
echo "== interference matrix, product build"
timeout -k 10 400 python tools/interfere.py 12 > $O/interfere.jsonl 2> $O/interfere.err; python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/diag10/interfere.jsonl")]
print("bad rounds total:", sum(r["bad_rounds"] for r in rows), "of", len(rows), "combinations")
PY
echo "== full GPU suite"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/t_full.log 2>&1; tail -5 $O/t_full.log
