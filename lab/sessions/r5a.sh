mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py -x -q > $O/parity.log 2>&1; echo rc=$? >> $O/parity.log
timeout 600 python tools/fused_error.py > $O/fused_error.json 2> $O/fused_error.err; echo rc=$? >> $O/fused_error.err
for i in 1 2; do
  timeout 400 python tools/mode_table.py --arith --outs f32 > $O/mode_paired_$i.json 2>> $O/mode.err
  GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_f32twice.so timeout 400 python tools/mode_table.py --arith --outs f32 > $O/mode_twice_$i.json 2>> $O/mode.err
done
timeout 300 python - > $O/ceiling.json 2> $O/ceiling.err <<'PY'
import json, torch, bench
from ggq_pkg import load_package
pkg = load_package()
print(json.dumps(bench.measured_ceiling(pkg, torch.device("cuda:0"))))
PY
tail -3 $O/parity.log; tail -2 $O/fused_error.err; cat $O/ceiling.json; tail -3 $O/mode.err
