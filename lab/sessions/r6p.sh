# round 6, session p: the 32-row kernel's LDS row pitch made odd for every format (r6o: Q5_0 at a 12-unit pitch 35-55 % slower than at 11): parity, then A/B
# in-tree (odd pitch) vs libggq_lab.so (built before the change: pitch = U units) vs libggq_oldalign.so (rounds 2-5: 32-element formats as aligned rows)
O=gpurun_out/r6p; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
  for v in intree lab oldalign; do
    lib=""; [ $v != intree ] && lib=$L/libggq_$v.so
    GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes 7296x2560,2432x2560,9728x2560,7296x2432,2432x2432 > $O/Q5_0_${v}_$i.json 2>> $O/err.log
  done
  for q in Q3_K Q6_K Q8_0 Q4_0 IQ4_XS Q4_1 Q5_1 Q2_K Q4_K; do
    for v in intree lab; do
      lib=""; [ $v != intree ] && lib=$L/libggq_$v.so
      GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype $q --m 32,64,128 --kernels mfma:0 --shapes 12288x3072,3072x12288,3072x3072 > $O/${q}_${v}_$i.json 2>> $O/err.log
    done
  done
done
python - <<'PY'
import json,glob
tab={}
for f in sorted(glob.glob("gpurun_out/r6p/*_?.json")):
    v=f.split("/")[-1][:-5]
    d=json.load(open(f))
    for r in d["rows"]: tab.setdefault((d["qtype"],r["weight"],r["m"]),{})[v.split("_",2)[-1] if v[0]!="I" else v.split("_",3)[-1]]=r.get("mfma:0")
for k,row in tab.items(): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6p/table.json","w"), indent=1)
PY
timeout 600 python tools/flux_forward_emulation.py --model sd35 --tokens 64 --fused-small-m --fused-mfma 256 --graph > $O/sd35_64.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/sd35_64.json')); print(d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['graph_replay_ms_per_step'])"
tail -5 $O/err.log
