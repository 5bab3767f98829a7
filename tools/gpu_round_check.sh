#!/bin/bash
# One GPU call's worth of checks while developing: the GPU tests, the contract line, the 16K intra-frame workload (single
# and, on one GPU, its banded form at world = 1), the decoder's rates.  Everything lands under gpurun_out/<name>/.
out=gpurun_out/${1:-check}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "bench done"
python bench.py --workload 16k_intra > $out/bench_16k.json 2> $out/bench_16k.err || { tail -5 $out/bench_16k.err; exit 1; }
python bench.py --workload 16k_intra --force-exchange --steps 50 --no-cpu-baseline > $out/bench_16k_banded1.json 2> $out/bench_16k_banded1.err || { tail -5 $out/bench_16k_banded1.err; exit 1; }
echo "16k done"
{ python tools/decode_bench.py --streams=3; python tools/decode_bench.py lossy --streams=3; python tools/decode_bench.py 4k --streams=3 --batch=4; } > $out/decode.txt 2>&1
cat $out/decode.txt
python tools/modes_time.py > $out/modes.txt 2>&1; tail -12 $out/modes.txt
