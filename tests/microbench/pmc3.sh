# Round 2: memory-system counters for the shipped Q4_K / Q2_K kernels next to the no-arithmetic streams (`ggq_microbench pmc3`).
# One rocprofv3 --pmc pass per counter set, --kernel-trace only (never combined with sys/hip/hsa tracing).
R=$PWD; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc3
rocprofv3 -L > $R/gpurun_out/pmc3/counters_available.txt 2>&1 || rocprofv3 --list-avail > $R/gpurun_out/pmc3/counters_available.txt 2>&1
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WR_UNCACHED_32B_sum" \
           "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/p$i -o p -- $R/tests/microbench/ggq_microbench pmc3 > $R/gpurun_out/pmc3/p$i.log 2>&1 || echo "pass $i ($SET) failed: $(tail -2 $R/gpurun_out/pmc3/p$i.log | tr '\n' ' ')"
done
ls $R/gpurun_out/pmc3/
