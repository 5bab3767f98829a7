#!/bin/bash
# repeated 9/7 decodes against the oracle under several band plans (tools/inv97_debug.py; run through gpurun)
for b in "32,32,8,4,4,4" "16,16,16,16,16,16" "32,32,32,32,32,32" "8,8,8,8,8,8" "4,4,4,4,4,4"; do
  echo "bands $b"; PICSONG_DWT_BANDS=$b python3 tools/inv97_debug.py 20 2>/dev/null | tail -1 | cut -c1-300
done
