R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_WRITEBACK_sum TCC_STREAMING_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_WRITE_sum TCC_READ_sum TCC_NORMAL_WRITEBACK_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/p$i -o p -- $R/tests/microbench/ggq_microbench pmc2 > $R/gpurun_out/pmc2/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmc2/p$i.log)"
done
ls $R/gpurun_out/pmc2/*/ | head -30
