#!/bin/bash
# Collects a round's judged measurements on the GPU box (run ONCE at the end of a round, from the repo root through
# gpurun: tools/collect_profiles.sh <name>; then tools/publish_profiles.sh <name> <tag> on the development box).
# Every rocprofv3 pass is its own run of bench.py --phase X: ONE launch shape per trace, so that a kernel's average in a
# kernel_stats.csv is that shape's (min ~ max) and every published fraction is one CSV row and one division.
# Counters (--pmc) in their own passes, never combined with sys / runtime traces.
set -e
out=gpurun_out/${1:-final}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the library every figure below belongs to: the binary's hash, and the hash of the sources it was built from
# (bench.py: source_hash(); a rebuild of the same sources changes the first, not the second)
{ sha256sum cuda-image-and-video-codec_amd/csrc/libpicsong_hip.so | cut -d' ' -f1; python3 -c "import importlib.util as u; s=u.spec_from_file_location('b','bench.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(m.source_hash())"; } > $out/library.sha256
trace() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 bench.py "$@" > $out/prof_$name.json 2> $out/prof_$name.err || { echo "trace $name FAILED"; tail -3 $out/prof_$name.err; }; echo "trace $name done"; }
pmc() { name=$1; ctr=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$name -- python3 bench.py "$@" > $out/pmc_$name.json 2> $out/pmc_$name.err || echo "pmc pass $name failed"; echo "pmc $name done"; }
# ---- kernel traces, one shape each
T="--steps 2 --warmup 1 --frames-per-step 36"
trace pipelined --phase pipelined $T                                   # 8K lossless, 3 streams x 3 frames per call (the headline's shape)
trace b3 --phase iso --batch 3 --steps 16                              # 3 frames per call, one stream, nothing else on the GPU
trace lone --phase lone --steps 16                                     # one frame per call
trace lossy_pipelined --phase pipelined $T --workload 8k_lossy
trace lossy_b6 --phase iso --workload 8k_lossy --steps 12              # 9/7 wl 6: six frames per call
trace lossy_lone --phase lone --workload 8k_lossy --steps 16
trace 4k_pipelined --phase pipelined --steps 2 --warmup 1 --frames-per-step 72 --workload 4k_lossless
trace 4k_b6 --phase iso --workload 4k_lossless --steps 12
trace 4k_lone --phase lone --workload 4k_lossless --steps 16
trace 16k --phase pipelined --workload 16k_intra --steps 8 --warmup 2  # config 5 on one GPU: one 16K x 16K frame per step
# ---- counters: FETCH_SIZE / WRITE_SIZE (HBM traffic, unit 1024 B; FETCH doubled on gfx950), SQ passes -- the LONE shape
# (a --pmc run serialises dispatches anyway), every workload
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
SQ2="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
for w in 8k_lossless 8k_lossy 4k_lossless; do
  P="--phase lone --steps 6 --pool 6 --workload $w"
  pmc fetch_$w FETCH_SIZE $P
  pmc write_$w WRITE_SIZE $P
  pmc sq_$w "$SQ1" $P
  pmc sq2_$w "$SQ2" $P
done
# rocprof's VALUBusy terms over the DEFAULT shape (dispatches serialised by the profiler: counters of lone kernels)
PP="--phase pipelined --steps 1 --warmup 1 --frames-per-step 36"
pmc sq_pipe "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" $PP
pmc sq_pipe2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES" $PP
make -C tools valu_probe > /dev/null          # (from tools/valu_probe.hip; __graft_entry__.build() builds it too)
timeout -k 10 240 tools/valu_probe $out/valu_probe.json > $out/valu_probe.txt 2>&1
echo "probe done"
# ---- the decoder: rates, kernel traces (lone frames), counters
{ python3 tools/decode_bench.py --streams=3; python3 tools/decode_bench.py lossy --streams=3; python3 tools/decode_bench.py 4k --streams=3 --batch=4; } > $out/decode.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_decode -- python3 tools/decode_bench.py > $out/prof_decode.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_decode_lossy -- python3 tools/decode_bench.py lossy > $out/prof_decode_lossy.log 2>&1
{ echo "# tools/pmc_decode.sh: SQ counters per dispatch (mean), tools/decode_bench.py 8K -type 0 wl 5"; tools/pmc_decode.sh ${1:-final}_l "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES";
  echo; echo "# the same, tools/decode_bench.py lossy: 8K -type 1 qs 0.5 wl 6"; tools/pmc_decode.sh ${1:-final}_y "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" lossy;
  echo; echo "# HBM traffic of the decode path (unit 1024 B; FETCH_SIZE doubled on gfx950), 8K -type 0 wl 5"; tools/pmc_decode.sh ${1:-final}_lf "FETCH_SIZE"; tools/pmc_decode.sh ${1:-final}_lw "WRITE_SIZE"; } > $out/pmc_decode.txt 2>&1
echo "decode done"
# ---- the other modes, a lone frame from Python, the fuzz run
python3 tools/modes_time.py > $out/modes_time.txt 2>&1 || true
python3 tools/lone_frame_time.py > $out/lone_frame.txt 2>&1 || true
python3 tools/rgb_probe.py > $out/rgb_probe.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rgb -- python3 tools/rgb_probe.py > $out/prof_rgb.log 2>&1 || true
python3 tools/fuzz_parity.py 120 4 > $out/fuzz_parity.txt 2>&1 || true
echo "modes done"
