#!/bin/bash
# A/B of the sizes' scan: scan_sizes_kernel launch (0) against the coder's last wave (1), same box: parity tests of the
# encode paths under the variant, then the bench lines (lone frame, three streams; 8K and 4K)
PICSONG_SCAN_IN_CODER=1 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "oracle or roundtrip or video or batched" 2>&1 | tail -2
for v in 0 1 0 1; do
  for w in 8k_lossless 4k_lossless; do
    PICSONG_SCAN_IN_CODER=$v python bench.py --steps 20 --no-cpu-baseline --no-b3 --workload $w > gpurun_out/abscan_${v}_$w.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/abscan_${v}_$w.json')); print('scan_in_coder=$v', '$w', d['value'], d['lone_frame']['ms'], d['lone_frame']['stage_ms'], d['timed_loop_outputs_ok'], d['roundtrip_ok'])"
  done
done
