#!/bin/bash
# rocprofv3 kernel averages of the 9/7 encode for 1 .. 6 frames per call (how the level launches amortise; run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 1 2 3 4 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bz_$b -- python3 bench.py --steps 4 --warmup 1 --frames-per-step 12 --streams 1 --batch $b --workload 8k_lossy --no-cpu-baseline --no-b3 > gpurun_out/bz_$b.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/bz_$b/*/*kernel_stats.csv")[0]
out=[]
for r in csv.DictReader(open(f)):
    if "dwt_fwd" in r["Name"] or "bpc_encode" in r["Name"]:
        out.append("%s %s x %.1f us (%.1f per frame)" % (r["Name"][14:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["AverageNs"])/1e3/$b))
print("batch $b:", "; ".join(out))
PY
done
