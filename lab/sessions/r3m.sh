#!/bin/bash
# round 3, GPU session M: timing ablations of the shared-tile GEMM (K-steps without the decode / without the MFMAs; results are wrong by
# construction, only the time matters): is a K-step the SUM of its decode and its MFMAs, or their maximum?
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
for rep in 1 2; do
for v in default nodecode nomma; do
  if [ $v = default ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  for wm in 2 4; do
  ( GGQ_TILE_WM=$wm timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072 --m 256,4608 --tiles 256 > $O/gemm_$v.wm$wm.$rep.json 2> $O/gemm_$v.wm$wm.$rep.err; echo "rc=$?" >> $O/gemm_$v.wm$wm.$rep.err )
  echo "== $v wm=$wm $rep"; grep -o '"m": [0-9]*\|"fused tile=256": [0-9.]*' $O/gemm_$v.wm$wm.$rep.err | paste - - 
  done
done; done
