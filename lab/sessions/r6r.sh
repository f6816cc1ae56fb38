# round 6, session r: every fused format x kernel x rows of x on one FLUX shape each way, looking for per-format cliffs like Q3_K's bank conflicts (r6p / r6q)
O=gpurun_out/r6r; mkdir -p $O
for q in Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K IQ4_NL IQ4_XS; do
  timeout 400 python tools/fused_sweep.py --qtype $q --m 1,4,8,16,32,64,128,256 --kernels small,mfma:16,mfma:0,mfma:256 --shapes 12288x3072,3072x12288 --reps 3 > $O/$q.json 2>> $O/err.log
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r6r/*.json")):
    d=json.load(open(f))
    for r in d["rows"]:
        print(d["qtype"], r["weight"], r["m"], {k:v for k,v in r.items() if k in ("small","mfma:16","mfma:0","mfma:256")})
PY
tail -3 $O/err.log
