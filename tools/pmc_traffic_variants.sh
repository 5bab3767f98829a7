#!/bin/bash
# usage: tools/pmc_traffic_variants.sh <variant> ...   -- the coder's FETCH_SIZE / WRITE_SIZE (one counter per run, a lone
# frame per call on one stream) for library variants (csrc/variants/<name>.so; "base" = the tree's library): which change
# moved the coder's HBM traffic.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
P="--steps 1 --warmup 1 --frames-per-step 4 --pool 4 --streams 1 --batch 1 --no-cpu-baseline --no-b3"
for n in "$@"; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  export PICSONG_SO=$so
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmcv_${n}_$c; rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py $P > /dev/null 2> $d.err || { echo "FAILED $n $c"; exit 1; }
    python3 tools/summarize_pmc.py $d/*/*counter_collection.csv | grep "bpc_encode" | sed "s/^/$n /"
  done
done
