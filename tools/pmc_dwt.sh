for v in lds leannoslp; do
  export PICSONG_SO=$GRAFT_REPO_ROOT/cuda-image-and-video-codec_amd/csrc/variants/$v.so
  echo "=== $v"
  tools/pmc_pass.sh ${v}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" --workload 8k_lossy --frames-per-step 4 --pool 4 --batch 1 | grep -A1 "dwt_fwd2"
  tools/pmc_pass.sh ${v}_b "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH" --workload 8k_lossy --frames-per-step 4 --pool 4 --batch 1 | grep -A1 "dwt_fwd2"
  tools/pmc_pass.sh ${v}_c "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" --workload 8k_lossy --frames-per-step 4 --pool 4 --batch 1 | grep -A1 "dwt_fwd2"
done
