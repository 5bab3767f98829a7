#!/bin/bash
# usage: tools/kmode_prof.sh <name> [k]  -- kernel times and SQ counters of the -k > 0 kernels (tools/kmode_probe.py)
name=$1; k=${2:-0.5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/km_$name -- python3 tools/kmode_probe.py $k > gpurun_out/km_$name.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --output-format csv -d gpurun_out/kmp_$name -- python3 tools/kmode_probe.py $k > gpurun_out/kmp_$name.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/km_$name/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "picsong" in r["Name"] and int(r["Calls"]) > 5:
        print("$name", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1000,2))
f=glob.glob("gpurun_out/kmp_$name/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "bpc_" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    print("    "+"  ".join("%s=%.4g"%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
