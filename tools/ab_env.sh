#!/bin/bash
# usage: tools/ab_env.sh "<ENV=val ...>" ...  -- the default bench under each environment setting, one summary line per run
for e in "$@"; do
 for w in ${AB_WORKLOADS:-8k_lossless 8k_lossy}; do
  tag=$(echo "$e$w" | tr -c 'A-Za-z0-9' '_')
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w ${AB_ARGS} > gpurun_out/abe_$tag.json 2> gpurun_out/abe_$tag.err || { echo "FAILED $e $w"; tail -3 gpurun_out/abe_$tag.err; continue; }
  python - <<PY
import json
d=json.load(open('gpurun_out/abe_$tag.json'))
print('%-28s %-12s %9.1f Mpix/s  ms/frame %.4f  dwt %.4f bpc %.4f | iso dwt %.4f bpc %.4f  ok %s' % ('$e', '$w', d['value'], d['ms_per_frame'], d['stage_ms']['dwt'], d['stage_ms']['bpc'], d['stage_ms_single_stream']['dwt'], d['stage_ms_single_stream']['bpc'], d['timed_loop_outputs_ok']))
PY
 done
done
