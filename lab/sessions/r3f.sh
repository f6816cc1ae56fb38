#!/bin/bash
# round 3, GPU session F: harness experiments (Q3_K line-exact groups, LDS vs no-LDS), the whole -m gpu suite on the tree with both store
# policies, bench.py with the three-view per_layer line
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
( timeout 300 tests/microbench/ggq_microbench abq3k > $O/microbench_q3k_line_exact.txt 2>&1 ); grep "^AB" $O/microbench_q3k_line_exact.txt | cut -c1-150
( timeout 300 tests/microbench/ggq_microbench ablds > $O/microbench_lds_vs_direct.txt 2>&1 ); grep "^AB" $O/microbench_lds_vs_direct.txt | cut -c1-150
( timeout 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic_source'][:80], d['cpu_baseline']['parity_vs_gpu'][:40], d['cpu_baseline'].get('parity_check_s'))
pl=d['workloads']['per_layer']
print(pl['value'], pl['roofline'], json.dumps(pl['config']['standalone_gpu_bound']), json.dumps(pl['config']['in_context']), pl['config']['eager_GBps'], pl['config']['eager_with_lookahead4'])
PY
for tok in 512 1024; do
  ( timeout 300 python tools/flux_forward_emulation.py --reps 5 --tokens $tok --fused-mfma 100000 > $O/emu_tok$tok.fm.json 2>> $O/emu.err ); python -c "
import json; d=json.load(open('$O/emu_tok$tok.fm.json')); print('tokens $tok fused-mfma auto', d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['dequant_cost_ms_per_step'])"
done
