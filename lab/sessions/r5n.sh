# round 5, session n: the MFMA K-split kernel's tile choice at 96 ... 256 rows (64- vs 128-row tiles vs auto), FLUX / T5 shapes (Q4_K) and SD3.5's 2432-column layers (Q5_0)
O=gpurun_out/r5n; mkdir -p $O
timeout 900 python tools/mfma_linear_bench.py --qtype Q4_K --m 96,128,192,256 --tiles 0,64,128 --shapes 3072x3072,12288x3072,3072x12288,21504x3072,4096x4096,10240x4096 > $O/q4k.json 2>> $O/err.log
timeout 600 python tools/mfma_linear_bench.py --qtype Q5_0 --m 96,128,192,256 --tiles 0,64,128 --shapes 2432x2432,7296x2432,9728x2432 > $O/q50.json 2>> $O/err.log
python - <<'PY'
import json
for f in ("q4k","q50"):
    d=json.load(open(f"gpurun_out/r5n/{f}.json"))
    for r in d["rows"]:
        print(f, r["weight"], r["m"], "def", r["dequant+F.linear"], "dense", r["F.linear dense-resident"], "auto", r.get("fused tile=auto"), "t64", r.get("fused tile=64"), "t128", r.get("fused tile=128"))
PY
