#!/bin/bash
# round 3, GPU session B: shared-tile GEMM after the loop restructuring (ping-pong) -- correctness, A/B of three builds, counters
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/test_mfma.log 2>&1; echo "rc=$?" >> $O/test_mfma.log ); tail -3 $O/test_mfma.log
for rep in 1 2; do
for v in default pp0 pp0s8; do
  if [ $v = default ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  ( timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288 --m 256,1024,4608 --tiles 256 > $O/gemm_$v.$rep.json 2> $O/gemm_$v.$rep.err; echo "rc=$?" >> $O/gemm_$v.$rep.err )
  echo "== $v $rep"; grep -o '"m": [0-9]*\|"fused tile=256": [0-9.]*\|"dequant+F.linear": [0-9.]*' $O/gemm_$v.$rep.err | paste - - - 
done; done
unset GGQ_HIP_LIB
bash tools/gemm_counters.sh r3b/pmc > $O/gemm_counters.txt 2>&1; cat $O/gemm_counters.txt | tail -45
