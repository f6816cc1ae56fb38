#!/bin/bash
# round 3, GPU session U: does the lookahead (several layers per launch) pay on models with SMALL layers (SD3.5-large, T5-xxl), where two or four dense weights still fit the Infinity Cache?
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3u; mkdir -p $O; : > $O/sweep.jsonl
run() { echo "{\"model\": \"$1\", \"tokens\": $2, \"mode\": \"$3\", \"result\":" >> $O/sweep.jsonl; timeout 300 python tools/flux_forward_emulation.py --model $1 --tokens $2 --reps 7 $3 >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl; }
for rep in 1 2; do
run sd35 4250 "--fused-small-m"
run sd35 4250 "--fused-small-m --lookahead 2"
run sd35 4250 "--fused-small-m --lookahead 4"
run t5 256 ""
run t5 256 "--lookahead 2"
run t5 256 "--lookahead 4"
run t5 256 "--lookahead 8"
done
python - <<'PY'
import json,re
s=open('gpurun_out/r3u/sweep.jsonl').read()
for m in re.finditer(r'\{"model": "(\w+)", "tokens": (\d+), "mode": "([^"]*)", "result":\s*(\{.*?\})\s*\}\n', s, re.S):
    try:
        r=json.loads(m.group(4)); print(m.group(1), m.group(2), m.group(3), r['ms_per_step_dequant_on_the_fly'], r['ms_per_step_dense_resident'], r['best_ms'])
    except Exception as e: print(m.group(1), m.group(2), m.group(3), 'ERR', e)
PY
tail -3 $O/sweep.err
