#!/bin/bash
# round 3, GPU session P: the bench line with the reference-on-this-GPU legs
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3p; mkdir -p $O
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err ); tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d.get('reference_on_this_gpu'))
pl=d['workloads']['per_layer']; print(pl['value'], pl['config']['in_context'], pl['config'].get('reference_on_this_gpu'))
PY
