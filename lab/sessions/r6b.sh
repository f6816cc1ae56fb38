# round 6, session b: the 16-row MFMA kernel (ggq_mfma16.hpp) -- parity first, then graph-replayed times against the shipped kernels, then the K-split width sweep (lab build)
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mfma.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -5 $O/tests.log
timeout 900 python tools/fused_sweep.py --m 1,4,8,16,32 --kernels small,mfma:0,mfma:16 > $O/sweep_main.json 2> $O/sweep_main.err; tail -40 $O/sweep_main.err
export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_lab.so
for kw in 2 4 6 8 12 16; do
  GGQ_MF16_KW=$kw timeout 600 python tools/fused_sweep.py --m 1,16,32 --kernels mfma:16 --shapes 12288x3072,18432x3072,3072x3072,3072x12288,21504x3072 > $O/sweep_kw$kw.json 2> $O/sweep_kw$kw.err
  echo kw=$kw; cat $O/sweep_kw$kw.err
done
