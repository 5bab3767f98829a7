#!/bin/bash
# usage: tools/prof_variant.sh <name> [bench args...]   (PICSONG_SO selects the library variant)
# rocprofv3 kernel stats of a short single-stream bench run; prints the DWT/BPC kernel averages.
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$name -- python3 bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline "$@" > gpurun_out/pv_$name.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pv_$name/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "picsong" in r["Name"]:
        print("$name", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1000,2))
PY
