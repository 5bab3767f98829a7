#!/bin/bash
# usage: tools/inv97_variants.sh <variant> ... -- rocprofv3 kernel averages of the 9/7 decode per library variant
# (csrc/variants/<name>.so) and band plan (INV97_BANDS: space-separated PICSONG_DWT_BANDS values)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants
for n in "$@"; do
 for b in ${INV97_BANDS:-16,8,4,4,4,4}; do
  export PICSONG_SO=$V/$n.so PICSONG_DWT_BANDS=$b
  rm -rf gpurun_out/i97v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/i97v -- python3 tools/decode_bench.py lossy > gpurun_out/i97v.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/i97v/*/*kernel_stats.csv"))[-1]
out=[]; tot=0
for r in csv.DictReader(open(f)):
    if "dwt_inv" in r["Name"]:
        out.append("%s %sx%.1f" % (r["Name"][31:54].replace(" ",""), int(r["Calls"])//23, float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3*(int(r["Calls"])//23)
print("$n $b: total %.1f us |" % tot, "; ".join(out))
PY
 done
done
rm -rf gpurun_out/i97v
