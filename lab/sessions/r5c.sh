O=gpurun_out/r5c; mkdir -p $O
L=$PWD/gpurun_tmp_libs
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py -x -q > $O/parity.log 2>&1; echo rc=$? >> $O/parity.log
for i in 1 2; do
  for v in main half pair0; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/mode_table.py --arith --outs f32 --steps 30 > $O/f32_${v}_$i.json 2>> $O/err.log
  done
done
unset GGQ_HIP_LIB
timeout 300 python - > $O/ceiling.json 2>> $O/err.log <<'PY'
import json, torch, bench
from ggq_pkg import load_package
pkg = load_package()
print(json.dumps(bench.measured_ceiling(pkg, torch.device("cuda:0"))))
PY
timeout 1200 python tools/token_sweep.py > $O/token_sweep.json 2>> $O/err.log
tail -2 $O/parity.log; cat $O/ceiling.json; cat $O/token_sweep.json; grep -v amdgpu.ids $O/err.log | tail -5
