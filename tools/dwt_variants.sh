#!/bin/bash
# usage: tools/dwt_variants.sh <variant> ...   -- rocprofv3 kernel averages of the 9/7 and 5/3 single-stream bench per
# library variant (csrc/variants/<name>.so), one summary line per DWT kernel
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
for n in "$@"; do
  for w in ${DV_WORKLOADS:-8k_lossy}; do
    PICSONG_SO=$V/$n.so tools/prof_variant.sh ${n}_$w --workload $w | grep -i "dwt\|bpc_encode" || { echo "FAILED $n $w"; tail -5 gpurun_out/pv_${n}_$w.log; }
  done
done
