# round 6, session x: the 32-row kernel with a span's x loads issued BEFORE the next span's weight prefetch (in-order vector memory: rounds 2-6 made every span wait for the
# HBM prefetch in front of its first MFMA) vs the old order (libggq_xlast.so = -DGGQ_MF_X_FIRST=0), every format
O=gpurun_out/r6x; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
SH=12288x3072,21504x3072,9216x3072,3072x3072,3072x12288
for i in 1 2; do
  for q in Q4_K Q5_K Q6_K Q3_K Q2_K Q8_0 Q4_0 Q4_1 Q5_0 Q5_1 IQ4_NL IQ4_XS; do
    for v in intree xlast; do
      lib=""; [ $v != intree ] && lib=$L/libggq_$v.so
      GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype $q --m 17,32,64,128,256 --kernels mfma:0 --shapes $SH --reps 3 > $O/${q}_${v}_$i.json 2>> $O/err.log
    done
  done
  for v in intree xlast; do
    lib=""; [ $v != intree ] && lib=$L/libggq_$v.so
    GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64,128 --kernels mfma:0 --shapes 7296x2432,2432x2432,9728x2432,2432x9728 --reps 3 > $O/sd35Q5_0_${v}_$i.json 2>> $O/err.log
  done
done
python - <<'PY'
import json,glob,os
tab={}
for f in sorted(glob.glob("gpurun_out/r6x/*_?.json")):
    d=json.load(open(f)); v=os.path.basename(f)[:-5].split("_")[-2]
    for r in d["rows"]:
        tab.setdefault((d["qtype"],r["weight"],r["m"]),{}).setdefault(v,[]).append(r.get("mfma:0"))
for k,row in sorted(tab.items()): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6x/table.json","w"), indent=1)
PY
tail -3 $O/err.log
