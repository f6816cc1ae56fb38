#!/bin/bash
# round 4, GPU session q (the round-3 sweep repeated on the final round-4 build: buffer stores, shared-scale fused linear): the emulated FLUX step over token counts (64 ... 4608) on the final build: default path, + fused 1-row layers, + fused MFMA kernel (small counts), wall and graph-replayed
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4q; mkdir -p $O; : > $O/sweep.jsonl
for T in 64 256 512 1024 2304 4608; do
  for MODE in "" "--fused-small-m"; do
    echo "{\"tokens\": $T, \"mode\": \"default $MODE\", \"result\":" >> $O/sweep.jsonl
    timeout 300 python tools/flux_forward_emulation.py --tokens $T --reps 5 --graph $MODE >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl
  done
  if [ $T -le 512 ]; then
    echo "{\"tokens\": $T, \"mode\": \"--fused-small-m --fused-mfma 256\", \"result\":" >> $O/sweep.jsonl
    timeout 300 python tools/flux_forward_emulation.py --tokens $T --reps 5 --graph --fused-small-m --fused-mfma 256 >> $O/sweep.jsonl 2>> $O/sweep.err; echo "}" >> $O/sweep.jsonl
  fi
done
python - <<'PY'
import json,re
s=open('gpurun_out/r4q/sweep.jsonl').read()
for m in re.finditer(r'\{"tokens": (\d+), "mode": "([^"]*)", "result":\s*(\{.*?\})\s*\}\n', s, re.S):
    try:
        r=json.loads(m.group(3))
        print(m.group(1), m.group(2), r['ms_per_step_dequant_on_the_fly'], r['ms_per_step_dense_resident'], r['graph_replay_ms_per_step'])
    except Exception as e: print(m.group(1), m.group(2), 'ERR', e)
PY
tail -3 $O/sweep.err
