#!/bin/bash
# round 3, GPU session C: shared-tile GEMM with LDS-DMA + 4-sub-phase ping-pong (A/B of three builds), lookahead (tests + emulated FLUX step)
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_lookahead.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -12 $O/tests.log
for rep in 1 2; do
for v in default pp1 nodma; do
  if [ $v = default ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  ( timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288 --m 256,1024,4608 --tiles 256 > $O/gemm_$v.$rep.json 2> $O/gemm_$v.$rep.err; echo "rc=$?" >> $O/gemm_$v.$rep.err )
  echo "== $v $rep"; grep -o '"m": [0-9]*\|"fused tile=256": [0-9.]*\|"dequant+F.linear": [0-9.]*' $O/gemm_$v.$rep.err | paste - - - 
done; done
unset GGQ_HIP_LIB
for la in 0 4 8 0 4 8; do
  ( timeout 300 python tools/flux_forward_emulation.py --reps 7 --lookahead $la > $O/emu_la$la.$RANDOM.json 2>> $O/emu.err )
done
cat $O/emu_la*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['lookahead'] and d['lookahead']['depth'], d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'], d['best_ms'])"
bash tools/gemm_counters.sh r3c/pmc > $O/gemm_counters.txt 2>&1; cat $O/gemm_counters.txt | tail -36
