#!/bin/bash
# round 3, GPU session R: fused 1-4-row linear with the LDS slice sized to the row (8 workgroups per CU instead of 4): tests, A/B by GGQ_LIN_PER_CU, in context
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3r; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_linear.py -x -q -m gpu > $O/tests_linear.log 2>&1; echo "rc=$?" >> $O/tests_linear.log ); tail -3 $O/tests_linear.log
for P in 4 8 4 8; do
  GGQ_LIN_PER_CU=$P timeout 300 python tools/fused_linear_bench.py Q4_K Q6_K > $O/bench_percu${P}_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_percu*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items(): print(' ',k, v['fused_us'], v['fused_graph_replay_us'], v['fused_graph_replay_packed_GBps'], v['linear_on_dense_resident_us'], v['dequant_plus_linear_us'])
PY
done
for P in 4 8 4 8; do
  echo "per_cu=$P"; GGQ_LIN_PER_CU=$P timeout 300 python tools/flux_forward_emulation.py --tokens 4608 --reps 7 --fused-small-m 2>>$O/emu.err | tee -a $O/emu_percu.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'])"
done
