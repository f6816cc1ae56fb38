#!/bin/bash
# round 4, session r: do the side-stream prefetch (overlap) and the lookahead pay at SMALL token counts, where the GEMMs are not bandwidth-bound? (both lost at 4608 tokens)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4r; mkdir -p $O; : > $O/sweep.jsonl
for T in 64 256 512; do
  for MODE in "--fused-small-m" "--fused-small-m --overlap" "--fused-small-m --lookahead 2" "--overlap"; do
    echo "{\"model\": \"flux\", \"tokens\": $T, \"mode\": \"$MODE\", \"result\":" >> $O/sweep.jsonl
    timeout 300 python tools/flux_forward_emulation.py --tokens $T --reps 5 $MODE >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl
  done
done
for T in 77 256; do
  for MODE in "" "--overlap" "--lookahead 2"; do
    echo "{\"model\": \"t5\", \"tokens\": $T, \"mode\": \"$MODE\", \"result\":" >> $O/sweep.jsonl
    timeout 300 python tools/flux_forward_emulation.py --model t5 --tokens $T --reps 5 $MODE >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl
  done
done
python - <<'PY'
import json,re
s=open('gpurun_out/r4r/sweep.jsonl').read()
for m in re.finditer(r'\{"model": "(\w+)", "tokens": (\d+), "mode": "([^"]*)", "result":\s*(\{.*?\})\s*\}\n', s, re.S):
    try:
        r=json.loads(m.group(4))
        print(m.group(1), m.group(2), m.group(3) or "default", r['ms_per_step_dequant_on_the_fly'], r['ms_per_step_dense_resident'])
    except Exception as e: print(m.group(1), m.group(2), m.group(3), 'ERR', e)
PY
tail -3 $O/sweep.err
