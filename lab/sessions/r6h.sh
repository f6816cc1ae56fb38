# round 6, session h: the whole GPU suite on the cleaned-up tree; bench --workload fused; fp16 vs bf16 activations (lever b); rocprofv3 kernel stats of the fused workload
O=gpurun_out/r6h; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -12 $O/tests.log
timeout 900 python bench.py --workload fused > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 1500 $O/bench_fused.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6h/bench_fused.json"))
for g in ("fused_small_m","fused_mfma"):
    for k,v in d["workloads"][g].items():
        if k!="how": print(g,k,v["ms_per_pass"],v["us_per_layer"],v["roofline"]["achieved"],v["roofline"]["frac"],v["TFLOPs"],v["layers_fused"],v["layers_declined"],v["parity_vs_fp64_on_oracle_weights"])
PY
for dt in bfloat16 float16; do timeout 600 python tools/fused_sweep.py --dtype $dt --m 1,4,8,32,64 --kernels small,mfma:0 --shapes 12288x3072,18432x3072,3072x12288 > $O/sweep_$dt.json 2> $O/sweep_$dt.err; done
python - <<'PY'
import json
for dt in ("bfloat16","float16"):
    for r in json.load(open(f"gpurun_out/r6h/sweep_{dt}.json"))["rows"]: print(dt, r["weight"], r["m"], r.get("small"), r.get("mfma:0"))
PY
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_fused -o fused -- python $GRAFT_REPO_ROOT/bench.py --workload fused > $GRAFT_REPO_ROOT/$O/prof_fused.log 2>&1; cd $GRAFT_REPO_ROOT; ls $O/prof_fused* | head; find $O/prof_fused -name "*kernel_stats*" | head -2 | xargs -I{} head -12 {}
