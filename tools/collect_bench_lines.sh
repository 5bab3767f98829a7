#!/bin/bash
# The bench lines of a round, run AFTER tools/publish_profiles.sh has put the counter summaries under profiles/ (so that the
# lines' offline fields -- traffic, valu_issue, valu_busy -- quote this library's counters): tools/collect_bench_lines.sh <name>
out=gpurun_out/${1:-final}
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench done"
python3 bench.py --no-cpu-baseline --workload 4k_lossless > $out/bench_4k.json 2> $out/bench_4k.err
python3 bench.py --no-cpu-baseline --workload 8k_lossy > $out/bench_8k_lossy.json 2> $out/bench_8k_lossy.err
python3 bench.py --workload 16k_intra > $out/bench_16k_intra.json 2> $out/bench_16k_intra.err
python3 bench.py --workload 16k_intra --force-exchange --steps 60 --no-cpu-baseline > $out/bench_16k_intra_banded_w1.json 2> $out/bench_16k_intra_banded_w1.err
python3 bench.py --force-exchange --steps 6 --no-cpu-baseline > $out/bench_exchange_w1.json 2> $out/bench_exchange_w1.err
echo "lines done"
