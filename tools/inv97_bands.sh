tools/pmc_decode.sh a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR" lossy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in "32,16,8,4,4,4" "32,8,4,4,4,4" "16,16,8,4,4,4" "16,8,8,8,4,4"; do
  export PICSONG_DWT_BANDS=$b
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inv97_b -- python3 tools/decode_bench.py lossy > gpurun_out/inv97_b.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/inv97_b/*/*kernel_stats.csv"))[-1]
out=[]
for r in csv.DictReader(open(f)):
    if "dwt_inv" in r["Name"]:
        out.append("%s %s x %.1f" % (r["Name"][31:41], r["Calls"], float(r["AverageNs"])/1e3))
print("bands $b:", "; ".join(out))
PY
  rm -rf gpurun_out/inv97_b
done
