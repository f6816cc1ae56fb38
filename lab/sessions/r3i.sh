#!/bin/bash
# round 3, GPU session I: the tree with sc1 stores in the single-tensor launches -- whole -m gpu suite, smoke, side-stream overlap revisited, bench.py
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
( timeout 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log ); tail -2 $O/smoke.log
for v in plain overlap plain overlap; do
  if [ $v = overlap ]; then F=--overlap; else F=; fi
  ( timeout 300 python tools/flux_forward_emulation.py --reps 7 $F > $O/emu_$v.$RANDOM.json 2>> $O/emu.err )
done
cat $O/emu_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('overlap' if d['overlap'] else 'default', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3i/bench.json'))
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['parity_vs_gpu'][:40])
pl=d['workloads']['per_layer']
print(pl['value'], json.dumps(pl['config']['standalone_gpu_bound']), json.dumps(pl['config']['in_context']), pl['config']['eager_GBps'])
PY
