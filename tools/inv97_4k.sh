for b in "" "8,4,4,4,4,4" "16,8,4,4,4,4" "16,4,4,4,4,4" "32,8,4,4,4,4"; do
  echo "4K lossy bands '${b:-rule}'"; PICSONG_DWT_BANDS=$b python3 tools/decode_bench.py lossy 4k --streams=3 --batch=4 2>/dev/null | grep -i "decode\|pipelined\|batch" | cut -c1-200
done
