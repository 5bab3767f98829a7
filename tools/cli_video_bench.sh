#!/bin/bash
# BASELINE config 4 through the C++ CLI, files on /tmp of the GPU box: 256 synthetic 4K frames,
# -type 0 wl 5, host file -> pinned memory -> H2D -> encode -> D2H -> file (PCIe-inclusive), then decode
# and compare.  usage: tools/cli_video_bench.sh [frames] [streams]
set -e
F=${1:-256}; NS=${2:-3}
cd $GRAFT_REPO_ROOT
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O
with open("/tmp/v4k.raw", "wb") as f:
    for i in range($F):
        f.write(O.gen_frame(3840, 2160, i).tobytes())
print("generated $F frames")
PY
BIN=cuda-image-and-video-codec_amd/host/PICSONG
LUT=tests/golden/lut/n1_lossless
$BIN -cd 0 -i /tmp/v4k.raw -o /tmp/v4k.enc -xSize 3840 -ySize 2160 -wl 5 -type 0 -video 1 -frames $F -numberOfStreams $NS -LUTFolder $LUT --metrics gpurun_out/cli_video_enc.json | tail -4
cat gpurun_out/cli_video_enc.json
$BIN -cd 1 -i /tmp/v4k.enc -o /tmp/v4k.dec -video 1 -numberOfStreams $NS -LUTFolder $LUT --metrics gpurun_out/cli_video_dec.json | tail -3
cat gpurun_out/cli_video_dec.json
cmp /tmp/v4k.raw /tmp/v4k.dec && echo "ROUNDTRIP IDENTICAL"
ls -la /tmp/v4k.raw /tmp/v4k.enc
