for v in cur nb16; do for w in 8k_lossless 8k_lossy; do
PICSONG_SO=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants/$v.so python bench.py --no-cpu-baseline --workload $w --streams 1 --batch 3 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $w b3', round(d['value']), d['stage_ms']['dwt'], d['roofline_dwt']['single_stream']['frac'])"
PICSONG_SO=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants/$v.so python bench.py --no-cpu-baseline --workload $w --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $w pipelined', round(d['value']), d['stage_ms_single_stream']['dwt'])"
done; done
