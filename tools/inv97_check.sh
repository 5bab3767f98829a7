#!/bin/bash
# GPU check of the 9/7 inverse kernels: the GPU suite (and once more with every wave replaying its band with true
# divisions), then the decoder's kernel trace and the pipelined decode rate with the lean kernel (default) and with
# dwt_inv_kernel's FAST instantiations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > gpurun_out/inv97_tests.log 2>&1 || { tail -30 gpurun_out/inv97_tests.log; exit 1; }
tail -2 gpurun_out/inv97_tests.log
PICSONG_DWT_EXACT_REPLAY=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or stage_by_stage or frame_codestream or batched_decode" > gpurun_out/inv97_tests_replay.log 2>&1 || { tail -30 gpurun_out/inv97_tests_replay.log; exit 1; }
tail -2 gpurun_out/inv97_tests_replay.log
for v in 1 0; do
  export PICSONG_DWT_INV97=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inv97_$v -- python3 tools/decode_bench.py lossy > gpurun_out/inv97_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/inv97_$v/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "dwt_inv" in r["Name"]:
        print("INV97=$v", r["Name"][:80], r["Calls"], "x %.1f us" % (float(r["AverageNs"])/1e3))
PY
  python3 tools/decode_bench.py lossy --streams=3 2>&1 | grep -i "decode\|pipelined"
done
