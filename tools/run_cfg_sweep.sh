#!/bin/bash
# sweep of bench call shapes (streams x frames per call, fused / two-launch DWT head) on the GPU box
run() { name=$1; shift; env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --steps 30 $ARGS > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err || { echo "FAILED $name"; return; }
  python -c "
import json; d=json.load(open('gpurun_out/cfg_$name.json'))
print('%-28s %9.1f Mpix/s  %.4f ms/frame  bpc %.4f dwt %.4f  iso bpc %.4f dwt %.4f  ok %s' % ('$name', d['value'], d['ms_per_frame'], d['stage_ms']['bpc'], d['stage_ms']['dwt'], d['stage_ms_single_stream']['bpc'], d['stage_ms_single_stream']['dwt'], d['timed_loop_outputs_ok'] and d['roundtrip_ok']))"; }
ARGS="--streams 2" run s2 A=1
ARGS="--streams 3" run s3 A=1
ARGS="--streams 4" run s4 A=1
ARGS="--streams 6" run s6 A=1
ARGS="--streams 3" run s3_twolaunch PICSONG_DWT_NOFUSE01=1
ARGS="--streams 5" run s5 A=1
ARGS="--streams 3 --batch 2" run s3_b2 A=1
ARGS="--streams 2 --batch 2" run s2_b2 A=1
ARGS="--streams 3 --workload 4k_lossless --batch 4" run 4k_s3_b4 A=1
ARGS="--streams 3 --workload 4k_lossless --batch 8" run 4k_s3_b8 A=1
ARGS="--streams 4 --workload 4k_lossless --batch 4" run 4k_s4_b4 A=1
ARGS="--streams 2 --workload 4k_lossless --batch 8" run 4k_s2_b8 A=1
