# PMC traffic of the whole-weight-set workloads (BASELINE configs[3] / configs[4]) measured on bench.py itself: FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 passes (--kernel-trace only), 3 plan launches each; tools/pmc_summarize.py --workloads adds them to profiles/pmc_traffic.json
# with the corrections calibrated by tests/microbench/pmc.sh (FETCH_SIZE x 2048, WRITE_SIZE x 1024).
R=$PWD; cd /tmp && export TMPDIR=/tmp
for W in flux sd35-t5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcw/$W-$C -o p -- python $R/bench.py --workload $W --steps 2 --warmup 1 --regions 1 --cpu-seconds 0 --no-ceiling > $R/gpurun_out/pmcw/$W-$C.log 2>&1 || echo "$W $C failed: $(tail -2 $R/gpurun_out/pmcw/$W-$C.log)"
  done
done
