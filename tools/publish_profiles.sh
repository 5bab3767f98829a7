#!/bin/bash
# usage: tools/publish_profiles.sh <gpurun_out subdir> <tag>   -- copies the summaries of a tools/collect_profiles.sh run
# (and, when present, of tools/collect_bench_lines.sh) into profiles/<tag>_* -- the files bench.py, DESIGN.md and
# profiles/README.md cite.  Run on the development box after the GPU call.
set -e
src=gpurun_out/$1; tag=$2; dst=profiles
ks() { ls $src/$1/*/*kernel_stats.csv 2>/dev/null | head -1; }
for n in pipelined b3 lone lossy_pipelined lossy_b6 lossy_lone 4k_pipelined 4k_b6 4k_lone 16k decode decode_lossy rgb; do
  f=$(ks prof_$n); [ -n "$f" ] && cp $f $dst/${tag}_kernel_stats_$n.csv
  [ -f $src/prof_$n.json ] && cp $src/prof_$n.json $dst/${tag}_kernel_stats_$n.line.json
done
cp $src/library.sha256 $dst/${tag}_library.sha256
cc() { ls $src/pmc_$1/*/*counter_collection.csv 2>/dev/null | head -1; }
lib="# libpicsong_hip.so sha256 $(head -1 $src/library.sha256), sources $(sed -n 2p $src/library.sha256)"
summ() { out=$1; shift; files=""; for n in "$@"; do f=$(cc $n); [ -n "$f" ] && files="$files $f"; done
  [ -z "$files" ] && return 0
  python3 tools/summarize_pmc.py $files > $out.tmp
  # keep the library's own kernels only (the bench also runs torch fills and copies), under a comment line that names
  # the library the counters were taken from (bench.py skips '#' lines)
  { echo "$lib"; head -1 $out.tmp; grep picsong $out.tmp || true; } > $out; rm -f $out.tmp; }
for w in 8k_lossless 8k_lossy 4k_lossless; do
  sfx="_$w"; [ $w = 8k_lossless ] && sfx=""
  summ $dst/${tag}_pmc_hbm$sfx.csv fetch_$w write_$w
  summ $dst/${tag}_pmc_sq$sfx.csv sq_$w sq2_$w
done
summ $dst/${tag}_pmc_sq_pipelined.csv sq_pipe sq_pipe2
for f in valu_probe.txt valu_probe.json decode.txt pmc_decode.txt modes_time.txt lone_frame.txt rgb_probe.txt fuzz_parity.txt; do [ -f $src/$f ] && cp $src/$f $dst/${tag}_$f; done
for f in bench bench_4k bench_8k_lossy bench_16k_intra bench_16k_intra_banded_w1 bench_exchange_w1; do [ -s $src/$f.json ] && cp $src/$f.json $dst/${tag}_$f.json; done
ls $dst | grep $tag
