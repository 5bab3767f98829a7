#!/bin/bash
# usage: tools/ab_decode.sh  -- the decoder's rates (tools/decode_bench.py: lone frame, three calls in flight; 8K 5/3, 8K 9/7,
# 4K four frames a call) under the environment switches that select the round-4 forms: 16-bit coefficients between decoder
# and synthesis (PICSONG_DEC_C16=0: the 32-bit arrays), synthesis levels 1 + 0 as one launch (PICSONG_DWT_NOFUSE_INV=1: two).
# AB_SO="name ..." also runs library variants csrc/variants/<name>.so in the default form.
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python tools/decode_bench.py --streams=3 2>/dev/null | grep decode; env "$@" python tools/decode_bench.py lossy --streams=3 2>/dev/null | grep decode; env "$@" python tools/decode_bench.py 4k --streams=3 --batch=4 2>/dev/null | grep "per call"; }
run "default (c16 + fused levels 1+0)" X=1
if [ -z "$AB_ONLY_SO" ]; then
run "c16, two launches" PICSONG_DWT_NOFUSE_INV=1
run "32-bit arrays (round 3 form)" PICSONG_DEC_C16=0
fi
for n in $AB_SO; do run "variant $n" PICSONG_SO=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants/$n.so; done
