#!/bin/bash
# round 3, GPU session N: the round's final measurements on the final tree -- GEMM sanity + table, PMC traffic (pool + workloads), rocprofv3
# kernel stats of bench.py, the bench line itself (with the PMC figures just collected), the whole -m gpu suite
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3n; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/tests_mfma.log 2>&1; echo "rc=$?" >> $O/tests_mfma.log ); tail -3 $O/tests_mfma.log
( timeout 400 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x3072,3072x12288,21504x3072 --m 64,256,1024,4608 --tiles 0,128,256 > $O/gemm_bench.json 2> $O/gemm_bench.err; echo "rc=$?" >> $O/gemm_bench.err ); tail -3 $O/gemm_bench.err | cut -c1-300
# PMC traffic: the pool through the harness, the whole-weight-set workloads through bench.py
bash tests/microbench/pmc.sh > $O/pmc_sh.log 2>&1
python tools/pmc_summarize.py gpurun_out/pmc --json profiles/pmc_traffic.json > $O/pmc_fetch_write_summary.txt 2> $O/pmc_summarize.err
mkdir -p gpurun_out/pmcw; bash tools/pmc_workloads.sh > $O/pmc_workloads.log 2>&1
python tools/pmc_summarize.py --workloads gpurun_out/pmcw profiles/pmc_traffic.json >> $O/pmc_fetch_write_summary.txt 2>> $O/pmc_summarize.err
cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -25 $O/pmc_fetch_write_summary.txt | cut -c1-200
# kernel stats of the bench process
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-per-qtype --no-per-mode --cpu-seconds 0 --no-workloads > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
ls $O/prof | head; 
# the bench line (profiles/pmc_traffic.json is now the one collected on this build)
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3n/bench.json'))
print(d['value'], d['roofline'], d['cpu_baseline']['value'], d['cpu_baseline']['parity_vs_gpu'][:60])
print({k:v['GB/s'] for k,v in d['per_qtype'].items()})
print({k:v['GB/s'] for k,v in d['per_mode'].items()})
for k,v in d['workloads'].items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('traffic'))
PY
( timeout 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
