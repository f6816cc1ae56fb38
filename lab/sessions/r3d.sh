#!/bin/bash
# round 3, GPU session D: the whole -m gpu suite, smoke, store-policy A/B in context, bench.py
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
( timeout 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -8 $O/tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log ); tail -2 $O/smoke.log
for rep in 1 2; do
for v in default plainst; do
  if [ $v = default ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_$v.so; fi
  ( timeout 300 python tools/flux_forward_emulation.py --reps 7 > $O/emu_$v.$rep.json 2>> $O/emu.err )
  python -c "
import json; d=json.load(open('$O/emu_$v.$rep.json')); print('$v', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
done; done
unset GGQ_HIP_LIB
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3d/bench.json'))
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['parity_vs_gpu'], d['cpu_baseline'].get('parity_check_s'))
print({k:v['GB/s'] for k,v in d['per_qtype'].items()})
print({k:v['GB/s'] for k,v in d['per_mode'].items()})
for k,v in d['workloads'].items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('cpu_baseline') or {}).get('parity_vs_gpu'), v.get('skipped'))
print(json.dumps(d['workloads']['per_layer']['config'])[:1500])
PY
