# round 6, session d: the 16-row kernel taken apart -- ablations (no decode / no x loads / neither), two spans of packed bytes in flight, no occupancy hint; two alternations
O=gpurun_out/r6d; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in lab pf2 wpe0 abl1 abl2 abl3; do
    GGQ_HIP_LIB=$L/libggq_$v.so timeout 300 python tools/fused_sweep.py --m 1,16,32 --kernels mfma:16 --shapes 12288x3072,18432x3072,3072x12288 > $O/${v}_$i.json 2>> $O/err.log
  done
done
python - <<'PY'
import json,glob
tab={}
for f in sorted(glob.glob("gpurun_out/r6d/*_?.json")):
    v=f.split("/")[-1][:-5]
    for r in json.load(open(f))["rows"]: tab.setdefault((r["weight"],r["m"]),{})[v]=r.get("mfma:16")
for k,row in tab.items(): print(k, row)
PY
