#!/bin/bash
# usage: tools/ab.sh <variant> [<variant> ...]   -- A/B of library variants (csrc/variants/<name>.so; "base" = the
# tree's libpicsong_hip.so) on the GPU box: the bench's value with 1 and 3 streams, one line per run.
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
W=${AB_WORKLOAD:-8k_lossless}
for n in "$@"; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  for st in ${AB_STREAMS:-1 3}; do
    PICSONG_SO=$so timeout -k 10 150 python bench.py --no-cpu-baseline --streams $st --workload $W ${AB_ARGS} > gpurun_out/ab_${n}_$st.json 2> gpurun_out/ab_${n}_$st.err || { echo "FAILED $n $st"; tail -3 gpurun_out/ab_${n}_$st.err; exit 1; }
    python - <<PY
import json
d=json.load(open('gpurun_out/ab_${n}_$st.json'))
print('%-10s streams %d  %9.1f Mpix/s  step %.4f ms  bpc %.4f  bpc_iso %.4f  dwt_iso %.4f  rt %s' % ('$n', $st, d['value'], d['ms_per_step'], d['stage_ms']['bpc'], d['stage_ms_single_stream']['bpc'], d['stage_ms_single_stream']['dwt'], d['roundtrip_ok']))
PY
  done
done
