#!/bin/bash
# bench.py stage times per frame for 3 .. 8 frames per picsong_encode_frames call on one stream (run through gpurun)
for w in 8k_lossy 8k_lossless; do for b in 3 4 6 8; do
  python3 bench.py --steps 6 --warmup 1 --frames-per-step 24 --streams 1 --batch $b --workload $w --no-cpu-baseline --no-b3 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', $b, j['value'], j['stage_ms'])"
done; done
