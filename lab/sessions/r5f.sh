# round 5, session f: A/B of the 0x64-byte-in-v_perm change on the fused 1-4-row linear (VALU-bound) and on the dequant pool (HBM-bound), two alternations
O=gpurun_out/r5f; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main permor; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/fused_linear_bench.py Q4_K Q5_K Q8_0 Q4_0 > $O/lin_${v}_$i.json 2>> $O/err.log
    timeout 300 python tools/mode_table.py --formats Q4_K,Q4_0,Q8_0 --outs f16,bf16 --steps 30 > $O/pool_${v}_$i.json 2>> $O/err.log
  done
done
cat $O/pool_*.json
