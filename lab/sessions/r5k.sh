#!/bin/bash
# round 5, session k: the judged measurements on the final dequant-kernel sources -- PMC traffic (microbench + the two weight-set workloads), rocprofv3 kernel stats of
# the bench command, the full bench line, the GPU suite, smoke().
set -u
R=$PWD; O=$R/gpurun_out/r5k; mkdir -p $O
bash tests/microbench/pmc.sh > $O/pmc.log 2>&1
mkdir -p gpurun_out/pmcw; bash tools/pmc_workloads.sh > $O/pmcw.log 2>&1
python tools/pmc_summarize.py gpurun_out/pmc --json $O/pmc_traffic.json > $O/pmc_fetch_write_summary.txt 2>$O/pmc_summarize.err
python tools/pmc_summarize.py --workloads gpurun_out/pmcw $O/pmc_traffic.json >> $O/pmc_fetch_write_summary.txt 2>>$O/pmc_summarize.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench below reports roofline.traffic for this very build
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-per-qtype --no-per-mode --cpu-seconds 0 --no-workloads > $O/bench_n1_under_rocprof.json 2> $O/bench_rocprof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
cd $R
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8) > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log; head -c 400 $O/bench_n1.json; echo; head -5 $O/bench_kernel_stats.csv | cut -c1-200; tail -8 $O/pmc_fetch_write_summary.txt | cut -c1-200
rm -rf $O/prof
