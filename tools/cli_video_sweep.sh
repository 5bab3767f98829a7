#!/bin/bash
# host-pipeline sweep of the CLI's video engines on the file of tools/cli_video_bench.sh (/tmp/v4k.raw, /tmp/v4k.enc):
# streams (pipeline slots) x reader / writer threads
cd $GRAFT_REPO_ROOT
BIN=cuda-image-and-video-codec_amd/host/PICSONG; LUT=tests/golden/lut/n1_lossless
for cfg in "3 2 2" "6 4 2" "6 5 4" "8 6 4" "12 8 6"; do
  set -- $cfg
  PICSONG_READERS=$2 PICSONG_WRITERS=$3 $BIN -cd 0 -i /tmp/v4k.raw -o /tmp/v4k_s.enc -xSize 3840 -ySize 2160 -wl 5 -type 0 -video 1 -frames 256 -numberOfStreams $1 -LUTFolder $LUT --metrics gpurun_out/cvs_e.json > /dev/null
  PICSONG_READERS=$2 PICSONG_WRITERS=$3 $BIN -cd 1 -i /tmp/v4k.enc -o /tmp/v4k_s.dec -video 1 -numberOfStreams $1 -LUTFolder $LUT --metrics gpurun_out/cvs_d.json > /dev/null
  python3 -c "
import json; e=json.load(open('gpurun_out/cvs_e.json')); d=json.load(open('gpurun_out/cvs_d.json'))
print('streams $1 readers $2 writers $3: encode %.1f Gpixel/s, decode %.1f Gpixel/s' % (e['mpixels_per_s']/1e3, d['mpixels_per_s']/1e3))"
  rm -f /tmp/v4k_s.enc /tmp/v4k_s.enc_SIZE /tmp/v4k_s.dec
done
