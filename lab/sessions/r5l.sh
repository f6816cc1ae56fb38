# round 5, session l: emulated steps on the final build -- FLUX again, SD3.5-large and the T5-xxl encoder: exact vs default vs dense-resident
O=gpurun_out/r5l; mkdir -p $O
timeout 1200 python tools/token_sweep.py --model flux --tokens 64,256,512,1024,2304,4608 > $O/flux.json 2>> $O/err.log
timeout 1200 python tools/token_sweep.py --model sd35 --tokens 64,256,1024,4250 > $O/sd35.json 2>> $O/err.log
timeout 900 python tools/token_sweep.py --model t5 --tokens 77,256,512 > $O/t5.json 2>> $O/err.log
python - <<'PY'
import json
for m in ("flux","sd35","t5"):
    d=json.load(open(f"gpurun_out/r5l/{m}.json"))
    print(m, {t:(r["exact_ms"],r["default_ms"],r["dense_resident_ms"]) for t,r in d["by_tokens"].items()})
PY
