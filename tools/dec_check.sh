#!/bin/bash
# usage (on the GPU box, from the repo root): tools/dec_check.sh <tag>  -- the decoder's round of checks: GPU tests, decode
# rates (5/3 and 9/7, a lone frame and three streams), SQ instruction counters and kernel-trace stats of a lone-frame run,
# and the encoder's bench line (the coders share their call sites and tables)
out=gpurun_out/$1; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
python tools/decode_bench.py --streams=3 > $out/decode.txt 2>&1
python tools/decode_bench.py lossy --streams=3 >> $out/decode.txt 2>&1
grep decode $out/decode.txt
tools/pmc_decode.sh $1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" > $out/pmc_decode.txt 2>&1; head -3 $out/pmc_decode.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_decode -- python3 tools/decode_bench.py > $out/prof_decode.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$out/prof_decode/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "picsong" in r["Name"]:
            print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python bench.py --steps 20 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['stage_ms_single_stream'])"
