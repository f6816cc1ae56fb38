#!/bin/bash
# round 3, GPU session T: the emulated step for the two halves of configs[4] (SD3.5-large MMDiT at 4250 tokens, T5-xxl encoder at 256 / 512 tokens) on the final build
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3t; mkdir -p $O; : > $O/sweep.jsonl
run() { echo "{\"model\": \"$1\", \"tokens\": $2, \"mode\": \"$3\", \"result\":" >> $O/sweep.jsonl; timeout 300 python tools/flux_forward_emulation.py --model $1 --tokens $2 --reps 5 --graph $3 >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl; }
run sd35 4250 ""
run sd35 4250 "--fused-small-m"
run sd35 1178 ""
run sd35 1178 "--fused-small-m"
run t5 256 ""
run t5 256 "--fused-mfma 256"
run t5 512 ""
run t5 77 ""
run t5 77 "--fused-mfma 256"
python - <<'PY'
import json,re
s=open('gpurun_out/r3t/sweep.jsonl').read()
for m in re.finditer(r'\{"model": "(\w+)", "tokens": (\d+), "mode": "([^"]*)", "result":\s*(\{.*?\})\s*\}\n', s, re.S):
    try:
        r=json.loads(m.group(4)); print(m.group(1), m.group(2), m.group(3), r['ms_per_step_dequant_on_the_fly'], r['ms_per_step_dense_resident'], r['graph_replay_ms_per_step'])
    except Exception as e: print(m.group(1), m.group(2), m.group(3), 'ERR', e)
PY
tail -3 $O/sweep.err
