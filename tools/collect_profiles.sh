#!/bin/bash
# Collects the judged measurements of a round on the GPU box (run from the repo root through gpurun):
# the default bench line, rocprofv3 kernel stats of the same command (shortened) and of a single-stream run,
# the 8K lossy and 4K variants, the FETCH_SIZE / WRITE_SIZE / SQ counter passes (each in its own run, never
# combined with sys/runtime traces), and the issue-rate probe.  tools/summarize_pmc.py turns the counter
# CSVs into the profiles/<tag>_pmc_*.csv summaries bench.py reads.
set -e
out=gpurun_out/${1:-final}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the library every figure below belongs to (bench.py prints the same hash in roofline.source)
sha256sum cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so | cut -d' ' -f1 > $out/library.sha256
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
S="--steps 2 --warmup 1 --frames-per-step 12 --no-cpu-baseline --no-b3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default -- python3 bench.py $S > $out/bench_prof_default.json 2> $out/prof_default.err
echo "prof default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_single -- python3 bench.py $S --streams 1 --batch 1 > $out/bench_prof_single.json 2> $out/prof_single.err
echo "prof single done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_lossy -- python3 bench.py $S --streams 1 --batch 1 --workload 8k_lossy > $out/bench_prof_lossy.json 2> $out/prof_lossy.err
echo "prof lossy done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_4k -- python3 bench.py $S --workload 4k_lossless > $out/bench_prof_4k.json 2> $out/prof_4k.err
echo "prof 4k done"
# the three-frames-per-call shape of the transform (picsong_encode_frames over three frames on one stream): kernel stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_b3 -- python3 bench.py $S --streams 1 --batch 3 > $out/bench_prof_b3.json 2> $out/prof_b3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_lossy_b3 -- python3 bench.py $S --streams 1 --batch 3 --workload 8k_lossy > $out/bench_prof_lossy_b3.json 2> $out/prof_lossy_b3.err
echo "prof b3 done"
P="--steps 1 --warmup 1 --frames-per-step 4 --pool 4 --streams 1 --batch 1 --no-cpu-baseline --no-b3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $P > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $P > $out/pmc_write.json 2> $out/pmc_write.err
echo "pmc write done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 bench.py $P > $out/pmc_sq.json 2> $out/pmc_sq.err
echo "pmc sq done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $out/pmc_sq2 -- python3 bench.py $P > $out/pmc_sq2.json 2> $out/pmc_sq2.err || echo "pmc sq2 pass failed (a counter of the list is not available on this box)"
echo "pmc sq2 done"
# the coder's bound as a measurement, in the DEFAULT shape (three streams, frames of three calls sharing the GPU)
PP="--steps 1 --warmup 1 --frames-per-step 48 --pool 16 --streams 3 --batch 1 --no-cpu-baseline --no-b3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc_sq_pipe -- python3 bench.py $PP > $out/pmc_sq_pipe.json 2> $out/pmc_sq_pipe.err || echo "pipelined sq pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/pmc_sq_pipe2 -- python3 bench.py $PP > $out/pmc_sq_pipe2.json 2> $out/pmc_sq_pipe2.err || echo "pipelined sq pass 2 failed"
echo "pmc pipelined done"
make -C tools valu_probe > /dev/null          # (from tools/valu_probe.hip; __graft_entry__.build() builds it too)
timeout -k 10 240 tools/valu_probe $out/valu_probe.json > $out/valu_probe.txt 2>&1
echo "probe done"
# the other workloads' bench lines, the three-frames-per-call shape, the decoder
python3 bench.py --no-cpu-baseline --workload 4k_lossless > $out/bench_4k.json 2> $out/bench_4k.err
python3 bench.py --no-cpu-baseline --workload 8k_lossy > $out/bench_8k_lossy.json 2> $out/bench_8k_lossy.err
python3 bench.py --no-cpu-baseline --streams 1 --batch 3 --steps 20 > $out/bench_8k_b3.json 2> $out/bench_8k_b3.err
python3 bench.py --no-cpu-baseline --streams 1 --batch 3 --steps 20 --workload 8k_lossy > $out/bench_8k_lossy_b3.json 2> $out/bench_8k_lossy_b3.err
echo "other benches done"
{ python3 tools/decode_bench.py --streams=3; python3 tools/decode_bench.py lossy --streams=3; python3 tools/decode_bench.py 4k --streams=3 --batch=4; } > $out/decode.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_decode -- python3 tools/decode_bench.py > $out/prof_decode.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_decode_lossy -- python3 tools/decode_bench.py lossy > $out/prof_decode_lossy.log 2>&1
{ echo "# tools/pmc_decode.sh: SQ counters per dispatch (mean), tools/decode_bench.py 8K -type 0 wl 5"; tools/pmc_decode.sh ${1:-final}_l "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES";
  echo; echo "# the same, tools/decode_bench.py lossy: 8K -type 1 qs 0.5 wl 6"; tools/pmc_decode.sh ${1:-final}_y "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" lossy; } > $out/pmc_decode.txt 2>&1
echo "decode done"
