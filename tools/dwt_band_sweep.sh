#!/bin/bash
# Sweep PICSONG_DWT_BANDS (rows per band, per level) on the 8K bench; prints the forward DWT stage time.
for b in ${SWEEP_BANDS:-"16,8,4,4,4" "32,8,4,4,4" "32,16,8,4,4" "16,16,8,4,4" "8,8,4,4,4" "32,8,8,8,8" "16,8,8,4,4" "32,16,16,8,8"}; do
  PICSONG_DWT_BANDS=$b python3 bench.py --no-cpu-baseline --streams 1 --steps 30 ${SWEEP_ARGS} | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$b', d['stage_ms'], d['ms_per_step'], d['roundtrip_ok'])" || exit 1
done
