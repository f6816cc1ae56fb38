# round 6, session i: the compile / autograd / doc tests with skip reasons; rocprofv3 kernel stats (csv) of bench --workload fused
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference.py tests/test_gpu_integration_doc.py tests/test_gpu_rows.py -q -rs -k "compile or autograd or documented or gather" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -40 $O/tests.log
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fused -o fused -- python $R/bench.py --workload fused > $R/$O/prof_fused.log 2>&1; cd $R; find $O/prof_fused -name "*stats*" | head; find $O/prof_fused -name "*kernel_stats.csv" | head -1 | xargs -I{} head -14 {}
