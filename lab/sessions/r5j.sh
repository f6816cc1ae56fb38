# round 5, session j: the MFMA K-split kernel's short last span (32-element blocks, cols % 64 == 0): tests, the fused-error table again, SD3.5 at 256 tokens
O=gpurun_out/r5j; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_fused_error.py tests/test_gpu_linear.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
timeout 600 python tools/fused_error.py > $O/fused_error.json 2>> $O/err.log; echo fe=$?
for t in 64 256; do
  timeout 300 python tools/flux_forward_emulation.py --model sd35 --tokens $t --reps 7 >> $O/sd35.jsonl 2>> $O/err.log
  timeout 300 python tools/flux_forward_emulation.py --model sd35 --tokens $t --reps 7 --fused-small-m --fused-mfma 256 >> $O/sd35.jsonl 2>> $O/err.log
done
cat $O/sd35.jsonl | cut -c1-330
