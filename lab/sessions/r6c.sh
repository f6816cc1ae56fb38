# round 6, session c: the 16-row kernel's variants -- k map (16 sub-block-per-lane / 17 x-contiguous) x blocks of W per wave (18 / 19 = two) -- parity, times, counters
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mfma.py -x -q -k "mfma16 or exact or short_last or randomized" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -5 $O/tests.log
timeout 1200 python tools/fused_sweep.py --m 1,4,8,16,32 --kernels small,mfma:0,mfma:16,mfma:17,mfma:18,mfma:19 --shapes 12288x3072,18432x3072,3072x12288,21504x3072,3072x3072 > $O/sweep.json 2> $O/sweep.err; cat $O/sweep.err
timeout 1200 bash tools/fused_counters.sh "mfma:16@18432x3072@1,small@18432x3072@1,mfma:16@12288x3072@16,mfma:17@12288x3072@16,mfma:19@12288x3072@16,mfma:0@12288x3072@32,mfma:19@12288x3072@32" _r6c > $O/counters.txt 2>&1; cat $O/counters.txt
