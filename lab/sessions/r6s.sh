# round 6, session s: the 16-row kernel without register spills for the 2-byte-aligned formats (one wave per SIMD less): A/B vs the uniform-occupancy build, and the
# K-split width under the new occupancy (lab build, GGQ_MF16_KW)
O=gpurun_out/r6s; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt
SH=12288x3072,3072x12288,3072x3072
for i in 1 2; do
  for q in Q8_0 Q5_0 Q4_0 IQ4_NL IQ4_XS Q6_K Q5_1; do
    GGQ_HIP_LIB= timeout 300 python tools/fused_sweep.py --qtype $q --m 1,4,8,16,32 --kernels mfma:16,mfma:32 --shapes $SH > $O/ab_${q}_intree_$i.json 2>> $O/err.log
    GGQ_HIP_LIB=$L/libggq_mf16uniformocc.so timeout 300 python tools/fused_sweep.py --qtype $q --m 1,4,8,16,32 --kernels mfma:16 --shapes $SH > $O/ab_${q}_uniform_$i.json 2>> $O/err.log
  done
done
for q in Q8_0 Q5_0 Q4_0; do
  for kw in 3 4 5 6 8; do
    GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF16_KW=$kw timeout 300 python tools/fused_sweep.py --qtype $q --m 1,8,16,32 --kernels mfma:16 --shapes $SH > $O/kw_${q}_kw$kw.json 2>> $O/err.log
  done
  GGQ_HIP_LIB=$L/libggq_lab.so timeout 300 python tools/fused_sweep.py --qtype $q --m 1,8,16,32 --kernels mfma:16 --shapes $SH > $O/kw_${q}_rule.json 2>> $O/err.log
done
python - <<'PY'
import json,glob,os
tab={}
for f in sorted(glob.glob("gpurun_out/r6s/*.json")):
    d=json.load(open(f)); v=os.path.basename(f)[:-5].replace(d["qtype"]+"_","")
    for r in d["rows"]:
        for k in ("mfma:16","mfma:32"):
            if k in r: tab.setdefault((d["qtype"],r["weight"],r["m"]),{})[v+("/32" if k=="mfma:32" else "")]=r[k]
for k,row in sorted(tab.items()): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6s/table.json","w"), indent=1)
PY
tail -3 $O/err.log
