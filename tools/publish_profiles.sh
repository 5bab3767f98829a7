#!/bin/bash
# usage: tools/publish_profiles.sh <gpurun_out subdir> <tag>   -- copies the summaries of a tools/collect_profiles.sh run
# into profiles/<tag>_* (the files bench.py and DESIGN.md cite).  Run on the development box after the GPU call.
set -e
src=gpurun_out/$1; tag=$2; dst=profiles
ks() { ls $src/$1/*/*kernel_stats.csv | head -1; }
cp $(ks prof_default) $dst/${tag}_kernel_stats.csv
cp $(ks prof_single) $dst/${tag}_kernel_stats_single_stream.csv
cp $(ks prof_lossy) $dst/${tag}_kernel_stats_8k_lossy.csv
cp $(ks prof_4k) $dst/${tag}_kernel_stats_4k.csv
[ -d $src/prof_decode ] && cp $(ks prof_decode) $dst/${tag}_kernel_stats_decode.csv
[ -d $src/prof_decode_lossy ] && cp $(ks prof_decode_lossy) $dst/${tag}_kernel_stats_decode_8k_lossy.csv
[ -d $src/prof_b3 ] && cp $(ks prof_b3) $dst/${tag}_kernel_stats_b3.csv
[ -d $src/prof_lossy_b3 ] && cp $(ks prof_lossy_b3) $dst/${tag}_kernel_stats_8k_lossy_b3.csv
cp $src/library.sha256 $dst/${tag}_library.sha256
[ -d $src/pmc_sq_pipe ] && python3 tools/summarize_pmc.py $src/pmc_sq_pipe/*/*counter_collection.csv $src/pmc_sq_pipe2/*/*counter_collection.csv > $dst/${tag}_pmc_sq_pipelined.csv
[ -f $src/pmc_decode.txt ] && cp $src/pmc_decode.txt $dst/${tag}_pmc_decode.txt
python3 tools/summarize_pmc.py $src/pmc_fetch/*/*counter_collection.csv $src/pmc_write/*/*counter_collection.csv > $dst/${tag}_pmc_hbm.csv
python3 tools/summarize_pmc.py $src/pmc_sq/*/*counter_collection.csv $src/pmc_sq2/*/*counter_collection.csv > $dst/${tag}_pmc_sq.csv
# keep the library's own kernels only (the bench also runs torch fills and copies)
# ... under a comment line that names the library the counters were taken from (bench.py skips '#' lines)
for f in $dst/${tag}_pmc_hbm.csv $dst/${tag}_pmc_sq.csv $dst/${tag}_pmc_sq_pipelined.csv; do [ -f $f ] || continue; { echo "# libpicsong_hip.so sha256 $(cat $src/library.sha256)"; head -1 $f; grep picsong $f; } > $f.tmp && mv $f.tmp $f; done
cp $src/valu_probe.txt $dst/${tag}_valu_probe.txt
cp $src/valu_probe.json $dst/${tag}_valu_probe.json
cp $src/bench.json $dst/${tag}_bench.json
for f in bench_4k bench_8k_lossy bench_8k_b3 bench_8k_lossy_b3; do [ -f $src/$f.json ] && cp $src/$f.json $dst/${tag}_$f.json; done
[ -f $src/decode.txt ] && cp $src/decode.txt $dst/${tag}_decode.txt
ls -la $dst | grep $tag
