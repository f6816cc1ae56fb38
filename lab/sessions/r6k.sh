# round 6, session k (lease A): compile tests again; legacy-format row threshold of the fused MFMA kernel; the 12 x 9 mode table with the shipped, all-workgroup-team and all-one-wave-team builds, two alternations
O=gpurun_out/r6k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reference.py -q -x -k "compile or autograd" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python tools/fused_sweep.py --qtype Q5_0 --m 32,64,96,128,192,256 --kernels default,mfma:0 --shapes 7296x2432,2432x2432,9728x2432,2432x9728 > $O/legacy_q50.json 2> $O/legacy_q50.err
timeout 900 python tools/fused_sweep.py --qtype Q8_0 --m 64,128,192,256 --kernels default,mfma:0 --shapes 12288x3072,3072x12288,4096x4096 > $O/legacy_q80.json 2> $O/legacy_q80.err
python - <<'PY'
import json
for f in ("legacy_q50","legacy_q80"):
    for r in json.load(open(f"gpurun_out/r6k/{f}.json"))["rows"]: print(f, r["weight"], r["m"], r.get("default"), r.get("mfma:0"))
PY
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in shipped coopall soloonly; do
    if [ $v = shipped ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 900 python tools/mode_table.py --arith > $O/mode_${v}_$i.json 2>> $O/mode_err.log
  done
done
unset GGQ_HIP_LIB
ls -la $O; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2 > $O/box.txt; cat $O/box.txt
