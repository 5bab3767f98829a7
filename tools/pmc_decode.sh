#!/bin/bash
# usage: tools/pmc_decode.sh <name> "<counters>" [decode_bench args] -- one rocprofv3 --pmc pass over tools/decode_bench.py
name=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmcd_$name -- python3 tools/decode_bench.py "$@" > gpurun_out/pmcd_$name.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcd_$name/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "picsong" in r["Kernel_Name"] and ("inv" in r["Kernel_Name"] or "decode" in r["Kernel_Name"]):
        acc[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    print("    "+"  ".join("%s=%.4g"%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
