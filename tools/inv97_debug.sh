V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
PICSONG_SO=$V/i97_fix2.so python3 tools/inv97_debug.py 100 2>/dev/null | tail -1 | cut -c1-300
export PICSONG_DWT_TRUST_STAGE=1
PICSONG_DWT_INV97=0 python3 tools/inv97_debug2.py save 2>/dev/null | tail -1
PICSONG_SO=$V/i97_fix2.so python3 tools/inv97_debug2.py cmp 100 2>/dev/null | tail -6 | cut -c1-300
