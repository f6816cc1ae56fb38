#!/bin/bash
# round 4, session t: the write-through store's cache policy IN CONTEXT (emulated FLUX step, 4608 tokens) for the aux combinations round 3 did not test
# (sc1|nt = 18, sc0|nt = 3, sc0|sc1|nt = 19) against the shipped sc1 (16); variant libraries -DGGQ_STORE_AUX=n, two alternations
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4t; mkdir -p $O; : > $O/ctx.jsonl
for i in 1 2; do
  for a in 16 18 3 19; do
    echo "{\"run\": $i, \"aux\": $a, \"result\":" >> $O/ctx.jsonl
    GGQ_HIP_LIB=$R/gpurun_tmp_libs/libggq_aux$a.so timeout 300 python tools/flux_forward_emulation.py --tokens 4608 --reps 5 >> $O/ctx.jsonl 2>> $O/err.txt; echo "}" >> $O/ctx.jsonl
    GGQ_HIP_LIB=$R/gpurun_tmp_libs/libggq_aux$a.so timeout 300 python tools/layer_latency.py > $O/lat_${a}_$i.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json,re
s=open('gpurun_out/r4t/ctx.jsonl').read()
for m in re.finditer(r'\{"run": (\d+), "aux": (\d+), "result":\s*(\{.*?\})\s*\}\n', s, re.S):
    r=json.loads(m.group(3)); print("run",m.group(1),"aux",m.group(2), r['ms_per_step_dequant_on_the_fly'], r['ms_per_step_dense_resident'], round(r['ms_per_step_dequant_on_the_fly']-r['ms_per_step_dense_resident'],3))
PY
tail -2 $O/err.txt
