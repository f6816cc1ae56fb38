#!/bin/bash
# round 3, GPU session G: K-step-64 shared-tile GEMM (Q4_K) -- correctness, A/B against the K-step-32 kernel in one library (GGQ_TILE64), counters
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -6 $O/tests.log
for rep in 1 2; do
for v in 1 0; do
  ( GGQ_TILE64=$v timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288,3072x3072,21504x3072 --m 256,1024,4608 --tiles 256 > $O/gemm_t64_$v.$rep.json 2> $O/gemm_t64_$v.$rep.err; echo "rc=$?" >> $O/gemm_t64_$v.$rep.err )
  echo "== tile64=$v $rep"; grep -o '"weight": "[0-9x]*"\|"m": [0-9]*\|"fused tile=256": [0-9.]*\|"dequant+F.linear": [0-9.]*\|"F.linear dense-resident": [0-9.]*' $O/gemm_t64_$v.$rep.err | paste - - - - -
done; done
bash tools/gemm_counters.sh r3g/pmc > $O/gemm_counters.txt 2>&1; cat $O/gemm_counters.txt | tail -36
