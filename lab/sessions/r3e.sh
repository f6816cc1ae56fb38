#!/bin/bash
# round 3, GPU session E: plain vs non-temporal stores at layer size (standalone per-layer latency AND in context), lookahead depths with
# plain stores, fused kernels in context, harness experiments (Q3_K line-exact groups, LDS vs no-LDS for Q4_0 / Q8_0)
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_overlap.py tests/test_gpu_reference.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -6 $O/tests.log
for rep in 1 2; do
for v in default ntst; do
  if [ $v = default ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  for cols in 3072 12288; do
    ( GGQ_LAYER_COLS=$cols timeout 200 python tools/layer_latency.py Q4_K Q5_K Q8_0 Q4_0 Q6_K > $O/layer_$v.$cols.$rep.json 2>> $O/layer.err )
    python -c "
import json; d=json.load(open('$O/layer_$v.$cols.$rep.json')); print('$v $cols', {k:(v['gpu_bound_us_per_call'], v['gpu_bound_GBps']) for k,v in d.items() if 'bfloat16' in k})"
  done
done; done
unset GGQ_HIP_LIB
for la in 0 2 4; do
  ( timeout 300 python tools/flux_forward_emulation.py --reps 7 --lookahead $la > $O/emu_la$la.json 2>> $O/emu.err )
  python -c "
import json; d=json.load(open('$O/emu_la$la.json')); print('lookahead $la', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
done
( timeout 300 python tools/flux_forward_emulation.py --reps 5 --fused-small-m > $O/emu_small.json 2>> $O/emu.err ); python -c "
import json; d=json.load(open('$O/emu_small.json')); print('fused-small-m', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
for tok in 128 512 1024; do
for fm in 0 100000; do
  ( timeout 300 python tools/flux_forward_emulation.py --reps 5 --tokens $tok --fused-mfma $fm > $O/emu_tok$tok.fm$fm.json 2>> $O/emu.err ); python -c "
import json; d=json.load(open('$O/emu_tok$tok.fm$fm.json')); print('tokens $tok fused-mfma $fm', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
done; done
( timeout 300 tests/microbench/ggq_microbench abq3k > $O/microbench_q3k_line_exact.txt 2>&1 ); grep "^AB" $O/microbench_q3k_line_exact.txt | cut -c1-150
( timeout 300 tests/microbench/ggq_microbench ablds > $O/microbench_lds_vs_direct.txt 2>&1 ); grep "^AB" $O/microbench_lds_vs_direct.txt | cut -c1-150
