# rocprofv3 counter passes over `ggq_microbench pmc` (calibration streams + the shipped kernels on the 64-pair pool)
R=$PWD; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -o p -- $R/tests/microbench/ggq_microbench pmc > $R/gpurun_out/pmc/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmc/p$i.log)"
done
grep "^PMC" $R/gpurun_out/pmc/p1.log
