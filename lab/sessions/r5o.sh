# round 5, session o: the auto policy's tall-weight rule (fused.AUTO_MAX_ROWS_TIMES_OUT): tests, the fused-error table, the FLUX sweep at the token counts it touches
O=gpurun_out/r5o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_fused_error.py tests/test_gpu_reference.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
timeout 600 python tools/fused_error.py > $O/fused_error.json 2>> $O/err.log; echo fe=$?
timeout 1200 python tools/token_sweep.py --model flux --tokens 64,256,512,1024,2304,4608 > $O/flux.json 2>> $O/err.log
timeout 900 python tools/token_sweep.py --model sd35 --tokens 64,256,1024,4250 > $O/sd35.json 2>> $O/err.log
python - <<'PY'
import json
for m in ("flux","sd35"):
    d=json.load(open(f"gpurun_out/r5o/{m}.json"))
    print(m, {t:(r["exact_ms"],r["default_ms"],r["dense_resident_ms"]) for t,r in d["by_tokens"].items()})
print(json.load(open("gpurun_out/r5o/fused_error.json"))["summary"])
PY
