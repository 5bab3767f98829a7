#!/bin/bash
# usage (GPU box, repo root): tools/ab_c16.sh  -- the bench and the decode bench with the encode frame paths' int16 coefficients
# off / on (PICSONG_C16=0 / 1), one summary line per run
for v in 0 1; do
  echo "== PICSONG_C16=$v"
  PICSONG_C16=$v python tools/decode_bench.py --streams=3 2>&1 | grep decode
  PICSONG_C16=$v python tools/decode_bench.py lossy --streams=3 2>&1 | grep decode
  for w in 8k_lossless 8k_lossy 4k_lossless; do
    PICSONG_C16=$v python bench.py --steps 20 --no-cpu-baseline --workload $w > gpurun_out/abc16_${v}_$w.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/abc16_${v}_$w.json')); print('$w', d['value'], d['lone_frame'], d['roofline_dwt']['single_stream']['frac'], d['roofline_dwt']['three_frames_per_call'])"
  done
done
