#!/bin/bash
# per-dispatch durations and gaps of one decoded frame's kernels (last frame of tools/decode_bench.py lossy) per band plan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in ${INV97_BANDS:-16,8,4,4,4,4}; do
  export PICSONG_DWT_BANDS=$b
  rm -rf gpurun_out/i97t
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/i97t -- python3 tools/decode_bench.py lossy > gpurun_out/i97t.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/i97t/*/*kernel_trace.csv"))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if "picsong" in r["Kernel_Name"]]
last=rows[-9:]
t0=int(last[0]["Start_Timestamp"])
print("bands $b")
prev=None
for r in last:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("  %-60s start %7.1f dur %6.1f gap %5.1f grid %s wg %s vgpr %s scratch %s" % (r["Kernel_Name"][9:69], (s-t0)/1e3, (e-s)/1e3, (s-prev)/1e3 if prev else 0, r.get("Grid_Size_X","?")+"x"+r.get("Grid_Size_Y","?"), r.get("Workgroup_Size_X","?"), r.get("VGPR_Count","?"), r.get("Scratch_Size", r.get("Private_Segment_Size","?"))))
    prev=e
PY
done
rm -rf gpurun_out/i97t
