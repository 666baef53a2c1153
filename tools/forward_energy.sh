#!/bin/bash
# energy of the forward transform (power x time per call of 250 clips, tools/power_clock.py) under the chunking switches:
# the step is bound by the socket's power limit, so what counts is joules per clip, not the stage's time alone
cd "$(dirname "$0")/.."
for cfg in "HPFW_FWD_CHUNK=16 HPFW_FWD_STREAMS=2" "HPFW_FWD_CHUNK=0" "HPFW_FWD_CHUNK=16 HPFW_FWD_STREAMS=1" "HPFW_FWD_CHUNK=32 HPFW_FWD_STREAMS=2" "HPFW_FWD_CHUNK=8 HPFW_FWD_STREAMS=2" "HPFW_FWD_CHUNK=16 HPFW_FWD_STREAMS=3"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/power_clock.py 1.5 2>/dev/null | grep -E "forward|whole" | awk '{ms=$0; sub(/ ms per call.*/,"",ms); n=split(ms,a," "); t=a[n]; p=$0; sub(/.*power +/,"",p); split(p,b," "); printf "%s   -> %.3f J per call\n", $0, t*b[1]/1000.0}'
done
