# round 5, session m: bf16 / fp32 arithmetic with 2-byte outputs: the shipped team table vs workgroup teams everywhere (-DGGQ_COOP_ALL_MODES), two alternations
O=gpurun_out/r5m; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main coopall; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 600 python tools/mode_table.py --arith --outs f16,bf16 --steps 30 > $O/m_${v}_$i.json 2>> $O/err.log
  done
done
