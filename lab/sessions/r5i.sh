# round 5, session i: fp32 output, bf16 / fp32 arithmetic of the K-quants: workgroup teams of 8192 elements (-DGGQ_F32_DOUBLE_GROUP) vs the shipped 4096, two alternations
O=gpurun_out/r5i; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main dbl; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/mode_table.py --arith --outs f32 --steps 30 --formats Q2_K,Q4_K,Q5_K,IQ4_XS > $O/f32_${v}_$i.json 2>> $O/err.log
  done
done
cat $O/f32_*.json
