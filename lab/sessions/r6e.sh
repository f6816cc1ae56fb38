# round 6, session e: the 32-row kernel with a launch-chosen K-split width (4..12 waves instead of always 4) -- parity, then the width sweep at 32 / 64 / 128 / 256 rows of x
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_linear.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -4 $O/tests.log
timeout 600 python tools/fused_sweep.py --m 16,32,64,128,256 --kernels mfma:0,mfma:32,mfma:64 --shapes 12288x3072,3072x12288,21504x3072,9216x3072,3072x3072 > $O/auto.json 2> $O/auto.err; cat $O/auto.err
export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_lab.so
for kw in 4 6 8 12; do
  GGQ_MF32_KW=$kw timeout 600 python tools/fused_sweep.py --m 32,64,128,256 --kernels mfma:32,mfma:64 --shapes 12288x3072,3072x12288,21504x3072 > $O/kw$kw.json 2> $O/kw$kw.err
done
python - <<'PY'
import json
tab={}
for kw in (4,6,8,12):
    for r in json.load(open(f"gpurun_out/r6e/kw{kw}.json"))["rows"]:
        for k in ("mfma:32","mfma:64"): tab.setdefault((r["weight"],r["m"],k),{})[kw]=r.get(k)
for k,row in tab.items(): print(k,row)
PY
