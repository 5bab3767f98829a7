#!/bin/bash
# usage: tools/prof_decode.sh <name> [decode_bench args...]: rocprofv3 kernel stats of tools/decode_bench.py
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd_$name -- python3 tools/decode_bench.py "$@" > gpurun_out/pd_$name.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pd_$name/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "picsong" in r["Name"] and int(r["Calls"]) > 5:
        print("$name", r["Name"][:78], r["Calls"], round(float(r["AverageNs"])/1000,2))
PY
