# round 6, session f: 16-row kernel with 4 / 2 blocks of W side by side in one workgroup (tile 20 / 21: x shared through L1) vs 1 (tile 16) vs the 32-row kernel
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mfma.py -x -q -k "mfma16 or exact or short_last or randomized" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python tools/fused_sweep.py --m 1,8,16,32 --kernels mfma:0,mfma:16,mfma:20,mfma:21 --shapes 12288x3072,18432x3072,3072x12288,21504x3072,3072x3072 > $O/sweep.json 2> $O/sweep.err; cat $O/sweep.err | cut -c1-400
export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_lab.so
for kw in 1 2 3 4; do
  GGQ_MF16_KW=$kw timeout 600 python tools/fused_sweep.py --m 16,32 --kernels mfma:20 --shapes 12288x3072,3072x12288,21504x3072 > $O/kw$kw.json 2> $O/kw$kw.err
done
python - <<'PY'
import json
tab={}
for kw in (1,2,3,4):
    for r in json.load(open(f"gpurun_out/r6f/kw{kw}.json"))["rows"]:
        tab.setdefault((r["weight"],r["m"]),{})[kw]=r.get("mfma:20")
for k,row in tab.items(): print(k,row)
PY
