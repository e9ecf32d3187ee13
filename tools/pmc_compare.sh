#!/bin/bash
# usage: tools/pmc_compare.sh "bench args A" "bench args B" ...  -> standard counter passes for each configuration
for cfg in "$@"; do
  echo "=== $cfg"
  BENCH_ARGS="$cfg" tools/pmc_gpu.sh "cmp" \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
    "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" || exit 1
  rm -rf gpurun_out/pmc_cmp
done
