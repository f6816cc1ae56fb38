#!/bin/bash
# round 6, session u (the r6m list again, after the LDS-pitch, spill and policy changes): the judged measurements on the final sources -- PMC traffic (microbench + the two weight-set workloads), rocprofv3 kernel stats of the bench command and of
# the fused workload, SQ / TCP / TCC counters of the final fused kernels, the full bench line, the fused-error table, the three token sweeps, the GPU suite, smoke().
set -u
R=$PWD; O=$R/gpurun_out/r6u; mkdir -p $O
bash tests/microbench/pmc.sh > $O/pmc.log 2>&1
mkdir -p gpurun_out/pmcw; bash tools/pmc_workloads.sh > $O/pmcw.log 2>&1
python tools/pmc_summarize.py gpurun_out/pmc --json $O/pmc_traffic.json > $O/pmc_fetch_write_summary.txt 2>$O/pmc_summarize.err
python tools/pmc_summarize.py --workloads gpurun_out/pmcw $O/pmc_traffic.json >> $O/pmc_fetch_write_summary.txt 2>>$O/pmc_summarize.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench below reports roofline.traffic for this very build
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-per-qtype --no-per-mode --cpu-seconds 0 --no-workloads > $O/bench_n1_under_rocprof.json 2> $O/bench_rocprof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused -o fused -- python $R/bench.py --workload fused > $O/bench_fused_under_rocprof.json 2> $O/bench_fused_rocprof.err)
find $O/prof_fused -name "*kernel_stats.csv" -exec cp {} $O/fused_kernel_stats.csv \;
cd $R
timeout 1500 bash tools/fused_counters.sh "small@18432x3072@1,small@9216x3072@1,mfma:0@18432x3072@1,mfma:0@12288x3072@4,mfma:0@12288x3072@8,mfma:0@3072x12288@8,mfma:0@12288x3072@32,mfma:0@12288x3072@64,mfma:0@3072x12288@64,mfma:0@12288x3072@256" _r6u > $O/fused_kernel_counters.txt 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 1500 python bench.py --workload fused-error > $O/fused_error.json 2> $O/fused_error.err
for M in flux sd35 t5; do timeout 2400 python tools/token_sweep.py --model $M > $O/token_sweep_$M.json 2> $O/token_sweep_$M.err; done
(timeout 2400 python -m pytest tests -q -m gpu -rfs -p no:cacheprovider 2>&1 | tail -25) > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.log 2>&1
grep -i "failed\|passed" $O/gpu_tests.log | tail -12; head -c 300 $O/bench_n1.json; echo; head -4 $O/bench_kernel_stats.csv | cut -c1-200; tail -6 $O/pmc_fetch_write_summary.txt | cut -c1-200; head -c 400 $O/fused_error.json; echo; for M in flux sd35 t5; do head -c 300 $O/token_sweep_$M.json; echo; done
rm -rf $O/prof $O/prof_fused $R/gpurun_out/pmc $R/gpurun_out/pmcw $R/gpurun_out/fusedpmc_r6u     # only the summaries travel back (gpurun merges at most 64 MiB)
du -sh $R/gpurun_out
