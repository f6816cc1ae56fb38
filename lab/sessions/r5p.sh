# round 5, session p: ggq_linear_small's per-row fetch with a wave-uniform unit count (shipped candidate) vs all six units predicated per lane (-DGGQ_LIN_ALL_UNITS), two alternations
O=gpurun_out/r5p; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main allunits; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/fused_linear_bench.py Q4_K Q5_K Q8_0 Q4_0 > $O/lin_${v}_$i.json 2>> $O/err.log
  done
done
unset GGQ_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_fused_error.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -2 $O/tests.log
