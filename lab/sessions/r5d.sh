O=gpurun_out/r5d; mkdir -p $O
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in main q3q6; do
    if [ $v = main ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 400 python tools/mode_table.py --arith --outs f32 --steps 30 --formats Q3_K,Q6_K > $O/f32_${v}_$i.json 2>> $O/err.log
  done
done
unset GGQ_HIP_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo rc=$? >> $O/gputests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err
tail -5 $O/gputests.log; tail -3 $O/bench.err; cat $O/f32_*.json; grep -v amdgpu.ids $O/err.log | tail -5
