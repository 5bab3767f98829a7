set -e
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > gpurun_out/dec_ring_tests.log 2>&1 || { tail -30 gpurun_out/dec_ring_tests.log; exit 1; }
tail -2 gpurun_out/dec_ring_tests.log
python3 tools/decode_bench.py --streams=3 2>/dev/null | grep -i "decode"
python3 tools/decode_bench.py lossy --streams=3 2>/dev/null | grep -i "decode"
python3 tools/decode_bench.py 4k --streams=3 --batch=4 2>/dev/null | grep -i "per call"
