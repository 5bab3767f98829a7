#!/bin/bash
# kernel-trace stats of the pipelined decode (three streams) with the decode path's coefficients as int16 / int32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export PICSONG_C16=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c16dec_$v -- python3 tools/decode_bench.py --streams=3 > gpurun_out/c16dec_$v.log 2>&1
  grep "decode" gpurun_out/c16dec_$v.log
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/c16dec_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "picsong" in r["Name"] and int(r["Calls"]) > 20:
            print("%-72s calls %5s avg %9.1f us  total %8.1f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
