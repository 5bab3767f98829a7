#!/bin/bash
# 8K decode throughput for several stream / frames-per-call shapes (tools/decode_bench.py; run through gpurun)
for cfg in "--streams=3 --batch=2" "--streams=2 --batch=2" "--streams=2 --batch=3" "--streams=4"; do
  echo "8K lossless $cfg"; python3 tools/decode_bench.py $cfg 2>/dev/null | grep -i "pipelined\|per call" | cut -c1-200
  echo "8K lossy $cfg"; python3 tools/decode_bench.py lossy $cfg 2>/dev/null | grep -i "pipelined\|per call" | cut -c1-200
done
