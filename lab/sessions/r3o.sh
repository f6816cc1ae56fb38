#!/bin/bash
# round 3, GPU session O: bench.py with the reference-on-this-GPU legs (headline pair + per-layer FLUX set), rocprofv3 kernel stats of the per-layer workload
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3o; mkdir -p $O
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d.get('reference_on_this_gpu'))
pl=d['workloads']['per_layer']; print(pl['value'], pl['config']['in_context'], pl['config'].get('reference_on_this_gpu'))
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o per_layer -- python $R/bench.py --workload per-layer > $O/per_layer_under_rocprof.json 2> $O/per_layer_under_rocprof.err )
ls $O/prof | head
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3o/prof/per_layer_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:110], r['Calls'], r['AverageNs'], r['Percentage'])
PY
