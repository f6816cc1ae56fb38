# round 6, session w: K split across workgroups with the slices added by the last workgroup of each tile (ticket counters: one launch instead of two).  Parity first, then the
# slice count forced in a lab build (GGQ_MF32_ZS) per shape and rows of x -- the question is whether evening out workgroups per CU (384 tiles on 256 CUs = 2 rounds) pays now
O=gpurun_out/r6w; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
SH=9216x3072,12288x3072,21504x3072,3072x3072,3072x12288,3072x15360,4096x4096,10240x4096,4096x10240
for i in 1 2; do
  for zs in 1 2 3 4 6; do
    GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF32_ZS=$zs timeout 600 python tools/fused_sweep.py --qtype Q4_K --m 32,64,128,256 --kernels mfma:0 --shapes $SH > $O/q4k_zs${zs}_$i.json 2>> $O/err.log
  done
  timeout 600 python tools/fused_sweep.py --qtype Q4_K --m 32,64,128,256 --kernels mfma:0 --shapes $SH > $O/q4k_rule_$i.json 2>> $O/err.log
done
for zs in 1 2 3 4; do
  GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF32_ZS=$zs timeout 600 python tools/fused_sweep.py --qtype Q5_0 --m 32,64,128 --kernels mfma:0 --shapes 7296x2432,2432x2432,9728x2432,2432x9728 > $O/q50_zs${zs}_1.json 2>> $O/err.log
done
python - <<'PY'
import json,glob,os
tab={}
for f in sorted(glob.glob("gpurun_out/r6w/q*_?.json")):
    d=json.load(open(f)); v=os.path.basename(f)[:-5].split("_")[1]
    for r in d["rows"]:
        tab.setdefault((d["qtype"],r["weight"],r["m"]),{}).setdefault(v,[]).append(r.get("mfma:0"))
for k,row in sorted(tab.items()): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6w/table.json","w"), indent=1)
PY
tail -3 $O/err.log
