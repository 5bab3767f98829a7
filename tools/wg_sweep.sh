#!/bin/bash
# workgroup-shape sweep of the coder kernels (variant libraries under csrc/variants/)
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
for n in e1 e2 base e8; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  for st in 1 3; do
    PICSONG_SO=$so timeout -k 10 120 python bench.py --no-cpu-baseline --streams $st > gpurun_out/wg_$n_$st.json || exit 1
    python -c "import json; d=json.load(open('gpurun_out/wg_$n_$st.json')); print('enc', '$n', 'streams', $st, d['value'], d['ms_per_step'])"
  done
done
for n in d1 d2 base d8; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  echo dec $n; PICSONG_SO=$so timeout -k 10 120 python tools/decode_bench.py --streams=2 || exit 1
  PICSONG_SO=$so timeout -k 10 120 python tools/decode_bench.py --streams=3 | tail -1 || exit 1
done
