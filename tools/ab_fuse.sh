for w in 8k_lossless 8k_lossy; do
for f in 0 1; do
  PICSONG_DWT_FUSE01=$f timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w > gpurun_out/fz2_${w}_$f.json 2> gpurun_out/fz2_${w}_$f.err || { echo FAILED $w $f; tail -3 gpurun_out/fz2_${w}_$f.err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/fz2_${w}_$f.json'))
print('$w fuse=$f %9.1f Mpix/s  ms/frame %.4f  dwt %.4f bpc %.4f | iso dwt %.4f bpc %.4f  ok %s' % (d['value'], d['ms_per_frame'], d['stage_ms']['dwt'], d['stage_ms']['bpc'], d['stage_ms_single_stream']['dwt'], d['stage_ms_single_stream']['bpc'], d['timed_loop_outputs_ok']))
PY
done
for b in 3 6; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --streams 1 --batch $b --steps 20 > gpurun_out/fz2_${w}_b$b.json 2> gpurun_out/fz2_${w}_b$b.err || { echo FAILED $w b$b; tail -3 gpurun_out/fz2_${w}_b$b.err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/fz2_${w}_b$b.json'))
print('$w batch=$b streams=1 %9.1f Mpix/s  ms/frame %.4f  dwt %.4f bpc %.4f | dwt roofline %s' % (d['value'], d['ms_per_frame'], d['stage_ms']['dwt'], d['stage_ms']['bpc'], json.dumps(d['roofline_dwt'].get('single_stream'))))
PY
done
done
