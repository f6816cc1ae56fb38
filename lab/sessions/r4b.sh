#!/bin/bash
# round 4, session b: (1) the LDS-tiled MFMA skeleton explorer (tests/microbench/gemm_skel.hip) at 4608 and 4096 rows; (2) the shipped shared-tile
# kernel's skeleton taken apart with the -DGGQ_GT_ABLATE=1/3/4/5 variant libraries, hipBLASLt beside it on the same box; (3) the new in-process
# multi-device tests.
set -u
O=gpurun_out/r4s2; mkdir -p $O
./tests/microbench/gemm_skel 4608 12288 3072 20 > $O/skel_4608x12288x3072.jsonl 2> $O/skel_4608.err
./tests/microbench/gemm_skel 4096 12288 3072 20 > $O/skel_4096x12288x3072.jsonl 2> $O/skel_4096.err
for v in base abl1 abl3 abl4 abl5; do
  if [ $v = base ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  python tools/mfma_linear_bench.py --shapes 12288x3072 --m 256,4608 --tiles 256 > $O/tile_$v.json 2> $O/tile_$v.err
done
unset GGQ_HIP_LIB
(timeout 900 python -m pytest tests/test_gpu_inproc.py -x -q 2>&1 | tail -15) > $O/inproc_tests.log
tail -3 $O/inproc_tests.log
cat $O/skel_4608x12288x3072.jsonl
