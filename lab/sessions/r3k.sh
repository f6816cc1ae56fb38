#!/bin/bash
# round 3, GPU session K: shared-tile GEMM with 16 waves (4 per SIMD, 64 x 64 per wave) vs 8 waves, one library (GGQ_TILE_WM)
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
for wm in 4 2; do
( GGQ_TILE_WM=$wm timeout 600 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/tests_wm$wm.log 2>&1; echo "rc=$?" >> $O/tests_wm$wm.log ); tail -3 $O/tests_wm$wm.log
done
for rep in 1 2; do
for wm in 4 2; do
  ( GGQ_TILE_WM=$wm timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288,3072x3072 --m 256,1024,4608 --tiles 256 > $O/gemm_wm$wm.$rep.json 2> $O/gemm_wm$wm.$rep.err; echo "rc=$?" >> $O/gemm_wm$wm.$rep.err )
  echo "== wm=$wm $rep"; grep -o '"weight": "[0-9x]*"\|"m": [0-9]*\|"fused tile=256": [0-9.]*\|"dequant+F.linear": [0-9.]*\|"F.linear dense-resident": [0-9.]*' $O/gemm_wm$wm.$rep.err | paste - - - - -
done; done
GGQ_TILE_WM=4 bash tools/gemm_counters.sh r3k/pmc > $O/gemm_counters_wm4.txt 2>&1; cat $O/gemm_counters_wm4.txt | tail -36
