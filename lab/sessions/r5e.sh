# round 5, session e: the 0x64-byte-in-v_perm change (ggq_device.hpp fields_h2): fused 1-4-row linear timing + every test that touches fp16 arithmetic
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python tools/fused_linear_bench.py Q4_K Q5_K Q8_0 > $O/fused_linear_small.json 2> $O/err.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_linear.py tests/test_gpu_mfma.py tests/test_gpu_rows.py tests/test_gpu_fused_error.py tests/test_gpu_reference.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -4 $O/tests.log; cat $O/fused_linear_small.json
