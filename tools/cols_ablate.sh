#!/bin/bash
cd "$(dirname "$0")/.."
for d in 0 1 2 4 3 7; do
  echo "== dbg=$d"
  HPFW_COLS_DBG=$d timeout -k 10 300 python tools/cols_stamps.py 2>/dev/null | tail -9
done
