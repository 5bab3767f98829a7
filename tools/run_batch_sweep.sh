set -x
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "batched" > gpurun_out/r2_tests4.log 2>&1; tail -5 gpurun_out/r2_tests4.log
for cfg in "8k_lossless 3 1" "8k_lossless 1 3" "8k_lossless 2 2" "8k_lossless 2 3" "4k_lossless 1 4" "4k_lossless 2 4" "4k_lossless 3 4" "4k_lossless 1 8" "4k_lossless 3 1"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $1 --streams $2 --batch $3 --steps 30 > gpurun_out/b3_$1_$2_$3.json 2> gpurun_out/b3_$1_$2_$3.err || { tail -5 gpurun_out/b3_$1_$2_$3.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/b3_$1_$2_$3.json'))
print('$1 streams $2 batch $3: %.1f Mpix/s  %.4f ms/frame  bpc %.4f iso %.4f dwt_iso %.4f loop_ok %s rt %s' % (d['value'], d['ms_per_frame'], d['stage_ms']['bpc'], d['stage_ms_single_stream']['bpc'], d['stage_ms_single_stream']['dwt'], d['timed_loop_outputs_ok'], d['roundtrip_ok']))"
done
