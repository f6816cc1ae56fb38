# round 6, session v: the 32-row kernel taken apart like the 16-row one was (R6-2): lab builds GGQ_MF_ABLATE = 1 no decode, 2 no global x loads, 3 both, 6 = no x at all (64+ rows: no LDS staging either), 7 = skeleton
O=gpurun_out/r6v; mkdir -p $O
L=$PWD/gpurun_tmp_libs
SH=12288x3072,21504x3072,3072x3072
for i in 1 2; do
  for v in intree 1 2 3 6 7; do
    lib=""; [ $v != intree ] && lib=$L/libggq_mfablate$v.so
    GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype Q4_K --m 32 --kernels mfma:32 --shapes $SH > $O/m32_${v}_$i.json 2>> $O/err.log
    GGQ_HIP_LIB=$lib timeout 300 python tools/fused_sweep.py --qtype Q4_K --m 64,128 --kernels mfma:64 --shapes $SH > $O/m64_${v}_$i.json 2>> $O/err.log
  done
done
python - <<'PY'
import json,glob,os
tab={}
for f in sorted(glob.glob("gpurun_out/r6v/m*_?.json")):
    d=json.load(open(f)); v=os.path.basename(f)[:-5].split("_")[1]
    for r in d["rows"]:
        tab.setdefault((r["weight"],r["m"]),{}).setdefault(v,[]).append(r.get("mfma:32", r.get("mfma:64")))
for k,row in sorted(tab.items()): print(k,row)
json.dump({str(k):v for k,v in tab.items()}, open("gpurun_out/r6v/table.json","w"), indent=1)
PY
tail -3 $O/err.log
