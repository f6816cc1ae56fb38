#!/bin/bash
# round 3, GPU session H: cache-policy bits of the single-tensor launches' stores (plain / sc1 / sc0 sc1 / sc0 builds, and non-temporal via
# GGQ_LAYER_NT_STORES=1): standalone per-layer latency and the emulated FLUX step
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
for rep in 1 2; do
for v in plain nt sc1 sc0sc1 sc0; do
  unset GGQ_HIP_LIB GGQ_LAYER_NT_STORES
  case $v in plain) ;; nt) export GGQ_LAYER_NT_STORES=1 ;; *) export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so ;; esac
  ( GGQ_LAYER_COLS=3072 timeout 200 python tools/layer_latency.py Q4_K Q8_0 > $O/layer_$v.3072.$rep.json 2>> $O/layer.err )
  ( GGQ_LAYER_COLS=12288 timeout 200 python tools/layer_latency.py Q4_K > $O/layer_$v.12288.$rep.json 2>> $O/layer.err )
  ( timeout 300 python tools/flux_forward_emulation.py --reps 5 > $O/emu_$v.$rep.json 2>> $O/emu.err )
  python -c "
import json
a=json.load(open('$O/layer_$v.3072.$rep.json')); b=json.load(open('$O/layer_$v.12288.$rep.json')); e=json.load(open('$O/emu_$v.$rep.json'))
print('$v', {k:v['gpu_bound_us_per_call'] for k,v in a.items() if 'bfloat16' in k}, {k:v['gpu_bound_us_per_call'] for k,v in b.items() if 'bfloat16' in k}, 'emu', e['ms_per_step_dequant_on_the_fly'], e['ms_per_step_dense_resident'], e['dequant_cost_ms_per_step'])"
done; done
