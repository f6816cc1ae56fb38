# round 6, session l (lease B): skip reasons of the compile tests; the mode table again on another lease (shipped, all-workgroup-team, all-one-wave-team builds, two alternations)
O=gpurun_out/r6l; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2 > $O/box.txt; cat $O/box.txt
timeout 900 python -m pytest tests/test_gpu_reference.py -q -rs -k "compile" 2>&1 | grep -v Warning > $O/tests.log; grep -A6 "SKIPPED" $O/tests.log | cut -c1-900 | head -40; tail -2 $O/tests.log
L=$PWD/gpurun_tmp_libs
for i in 1 2; do
  for v in shipped coopall soloonly; do
    if [ $v = shipped ]; then unset GGQ_HIP_LIB; else export GGQ_HIP_LIB=$L/libggq_$v.so; fi
    timeout 900 python tools/mode_table.py --arith > $O/mode_${v}_$i.json 2>> $O/mode_err.log
  done
done
unset GGQ_HIP_LIB
