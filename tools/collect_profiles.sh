#!/bin/bash
# Collects the judged measurements of a round on the GPU box (run from the repo root through gpurun):
# the default bench line, rocprofv3 kernel stats of the same command and of a single-stream run,
# the 8K lossy variant, and the FETCH_SIZE / WRITE_SIZE / SQ counter passes (each in its own run,
# never combined with sys/runtime traces).
set -e
out=gpurun_out/${1:-final}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default -- python3 bench.py --no-cpu-baseline > $out/bench_prof_default.json 2> $out/prof_default.err
echo "prof default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_single -- python3 bench.py --steps 30 --warmup 3 --streams 1 --no-cpu-baseline > $out/bench_prof_single.json 2> $out/prof_single.err
echo "prof single done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_lossy -- python3 bench.py --steps 30 --warmup 3 --streams 1 --workload 8k_lossy --no-cpu-baseline > $out/bench_prof_lossy.json 2> $out/prof_lossy.err
echo "prof lossy done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
echo "pmc write done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > $out/pmc_sq.json 2> $out/pmc_sq.err
echo "pmc sq done"
