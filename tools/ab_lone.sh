#!/bin/bash
# usage: tools/ab_lone.sh <variant> ...  -- a lone frame (bench.py --phase lone: one frame per call, one stream) and the
# pipelined loop (--phase pipelined) for library variants (csrc/variants/<name>.so; "base" = the tree's), 8K and 4K
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
for n in "$@"; do
  so=$V/$n.so; [ $n = base ] && so=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so
  for w in 8k_lossless 4k_lossless; do
    l=$(PICSONG_SO=$so python bench.py --phase lone --steps 30 --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_frame']; print('lone %.4f ms (dwt %.4f bpc %.4f pack %.4f) = %.1f Gpx/s' % (sum(s.values()), s['dwt'], s['bpc'], s['pack'], d['mpixels_per_s']/1e3))")
    p=$(PICSONG_SO=$so python bench.py --phase pipelined --steps 8 --warmup 2 --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined %.1f Gpx/s' % (d['value']/1e3))")
    echo "$n $w: $l; $p"
  done
done
