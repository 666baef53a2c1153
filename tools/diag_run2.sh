#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/diag2
mkdir -p $O
echo "== forward transform under switches, one process"
timeout -k 10 500 python tools/fwd_diag.py 3 30.0 128 > $O/fwd.jsonl 2> $O/fwd.err
tail -3 $O/fwd.err
python - <<'PY'
import json
for ln in open("gpurun_out/diag2/fwd.jsonl"):
    d = json.loads(ln)
    print(d["config"], "| x bad clips per rep:", [len(b) for b in d["x_bad_clips"]], "| z records:", len(d["z"]), d["z"][:2], d.get("x_detail", [])[:2])
PY
echo "== two processes, 5 s clips, started together"
rm -f $O/start5*
(timeout -k 10 400 python tools/shared_gpu_diag.py 4000 250 5.0 120 $PWD/$O/start5 > $O/diag5_a.json 2> $O/diag5_a.err) &
timeout -k 10 400 python tools/shared_gpu_diag.py 4100 250 5.0 120 $PWD/$O/start5 > $O/diag5_b.json 2> $O/diag5_b.err; wait
python - <<'PY'
import json
for f in ("a", "b"):
    try:
        d = json.load(open(f"gpurun_out/diag2/diag5_{f}.json"))
        print(f, d["elapsed_s"], d["totals"], d["n_events"])
    except Exception as e:
        print(f, "no result", e)
PY
echo "== two processes, 30 s clips, started together"
rm -f $O/start30*
(timeout -k 10 500 python tools/shared_gpu_diag.py 4000 40 30.0 128 $PWD/$O/start30 > $O/diag30_a.json 2> $O/diag30_a.err) &
timeout -k 10 500 python tools/shared_gpu_diag.py 4100 40 30.0 128 $PWD/$O/start30 > $O/diag30_b.json 2> $O/diag30_b.err; wait
python - <<'PY'
import json
for f in ("a", "b"):
    try:
        d = json.load(open(f"gpurun_out/diag2/diag30_{f}.json"))
        print(f, d["elapsed_s"], d["totals"], d["n_events"])
    except Exception as e:
        print(f, "no result", e)
PY
