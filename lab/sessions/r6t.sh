# round 6, session t: the row threshold of the fused MFMA path for the 32-element-block formats again (r6k measured it on the bank-conflicted 32-row kernel), and Q3_K / Q6_K
O=gpurun_out/r6t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
timeout 900 python tools/fused_sweep.py --qtype Q5_0 --m 32,64,96,128,192,256 --kernels default,mfma:0 --shapes 7296x2432,2432x2432,9728x2432,2432x9728 > $O/legacy_Q5_0_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q5_1 --m 64,96,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q5_1_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q8_0 --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q8_0_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q4_0 --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q4_0_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q3_K --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q3_K_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q6_K --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q6_K_$i.json 2>> $O/err.log
timeout 900 python tools/fused_sweep.py --qtype Q4_K --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_Q4_K_$i.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
tab={}
for f in sorted(glob.glob("gpurun_out/r6t/legacy_*.json")):
    d=json.load(open(f))
    for r in d["rows"]:
        t=tab.setdefault((d["qtype"],r["weight"],r["m"]),{"default":[],"fused":[]})
        t["default"].append(r.get("default")); t["fused"].append(r.get("mfma:0"))
for k,v in sorted(tab.items()): print(k, v)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6t/table.json","w"), indent=1)
PY
tail -3 $O/err.log
