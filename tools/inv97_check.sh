#!/bin/bash
# GPU check of the 9/7 inverse kernels: parity tests that touch them, then the decoder's kernel trace and
# the pipelined decode rate with the lean kernel (default) and with dwt_inv_kernel's FAST instantiations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q -k "lossy or 97 or dwt or psnr or inverse or decode" > gpurun_out/inv97_tests.log 2>&1 || { tail -30 gpurun_out/inv97_tests.log; exit 1; }
tail -2 gpurun_out/inv97_tests.log
for v in 1 0; do
  export PICSONG_DWT_INV97=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inv97_$v -- python3 tools/decode_bench.py lossy > gpurun_out/inv97_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/inv97_$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "dwt_inv" in r["Name"]:
        print("INV97=$v", r["Name"][:70], r["Calls"], "x %.1f us" % (float(r["AverageNs"])/1e3))
PY
  python3 tools/decode_bench.py lossy --streams=3 2>&1 | grep -i "decode\|pipelined"
done
