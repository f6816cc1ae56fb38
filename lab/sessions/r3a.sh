#!/bin/bash
# round 3, GPU session A: any-order probe, new parity tests, shared-tile GEMM correctness + first timing, bench.py
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
( timeout 60 tools/probes/anyorder_probe > $O/anyorder.txt 2>&1; echo "rc=$?" >> $O/anyorder.txt )
( timeout 900 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/test_mfma.log 2>&1; echo "rc=$?" >> $O/test_mfma.log )
tail -5 $O/test_mfma.log
( timeout 400 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x3072,3072x12288 --m 256,512,1024,4608 --tiles 128,256 > $O/gemm_bench.json 2> $O/gemm_bench.err; echo "rc=$?" >> $O/gemm_bench.err )
tail -14 $O/gemm_bench.err
( timeout 900 python -m pytest tests/test_gpu_fullconfig.py "tests/test_gpu_reference.py::test_reference_torch_ops_on_the_gpu_at_flux_sizes" -x -q -m gpu > $O/test_full.log 2>&1; echo "rc=$?" >> $O/test_full.log )
tail -5 $O/test_full.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -3 $O/bench.err; head -c 1500 $O/bench.json
cat $O/anyorder.txt
