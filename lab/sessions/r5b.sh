O=gpurun_out/r5b; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main pair0 pair1 pair1coop pair2coop; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/mode_table.py --arith --outs f32 --steps 30 > $O/f32_${v}_$i.json 2>> $O/err.log
  done
done
for i in 1 2; do
  for v in main q2g32x0 q2g32x4 q2ntl; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 300 python tools/mode_table.py --formats Q2_K --outs f16,bf16 --steps 40 > $O/q2_${v}_$i.json 2>> $O/err.log
  done
done
unset GGQ_HIP_LIB
timeout 300 python - > $O/ceiling.json 2>> $O/err.log <<'PY'
import json, torch, bench
from ggq_pkg import load_package
pkg = load_package()
print(json.dumps(bench.measured_ceiling(pkg, torch.device("cuda:0"))))
PY
cat $O/ceiling.json; grep -v amdgpu.ids $O/err.log | tail -5
