# round 6, session q: (1) 32-row kernel with the linear stash kept where the pitch was odd already (Q4_K / Q5_K / Q4_1 / Q5_1: r6p showed +2-5 % at 128 rows from the
# stash address registers); (2) the 256 x 256 tile kernel's staging pitch made odd (unit count | 1) vs the even-pitch A/B build
O=gpurun_out/r6q; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 1200 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_linear.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
  for q in Q4_K Q5_1 Q5_K; do
    GGQ_HIP_LIB= timeout 300 python tools/fused_sweep.py --qtype $q --m 32,64,128 --kernels mfma:0 --shapes 12288x3072,3072x12288,3072x3072 > $O/m32_${q}_intree_$i.json 2>> $O/err.log
    GGQ_HIP_LIB=$L/libggq_lab.so timeout 300 python tools/fused_sweep.py --qtype $q --m 32,64,128 --kernels mfma:0 --shapes 12288x3072,3072x12288,3072x3072 > $O/m32_${q}_lab_$i.json 2>> $O/err.log
  done
  for q in Q3_K Q6_K Q8_0 Q4_0 Q5_0 Q5_1 Q4_K IQ4_XS Q2_K; do
    for v in intree gtevenpitch; do
      lib=""; [ $v != intree ] && lib=$L/libggq_$v.so
      GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype $q --m 256,1024 --kernels mfma:256 --shapes 12288x3072,3072x12288 > $O/tile_${q}_${v}_$i.json 2>> $O/err.log
    done
  done
done
python - <<'PY'
import json,glob,os
tab={}
for f in sorted(glob.glob("gpurun_out/r6q/*_?.json")):
    d=json.load(open(f)); v=os.path.basename(f)[:-5].replace(d["qtype"]+"_","")
    for r in d["rows"]: tab.setdefault((d["qtype"],r["weight"],r["m"]),{})[v]=r.get("mfma:0", r.get("mfma:256"))
for k,row in tab.items(): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6q/table.json","w"), indent=1)
PY
tail -5 $O/err.log
