#!/bin/bash
# usage: tools/ab_decode.sh <variant> [<variant> ...]  -- A/B of library variants (csrc/variants/<name>.so; "base" = the
# tree's libpicsong_hip.so) on the GPU box: tools/decode_bench.py (a lone 8K frame and three streams), one block per variant.
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
for n in "$@"; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  echo "== $n"
  PICSONG_SO=$so timeout -k 10 200 python tools/decode_bench.py ${AB_ARGS:---streams=3} 2>&1 | grep decode || { echo "FAILED $n"; exit 1; }
done
