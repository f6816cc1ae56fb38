#!/bin/bash
# round 3, GPU session AF: Q4_K / Q5_K shared per-span scale decode in the shared-tile GEMM: tests (bit-identical weights), then A/B against a build without it, two alternations
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3af; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu > $O/tests_mfma.log 2>&1; echo "rc=$?" >> $O/tests_mfma.log ); tail -3 $O/tests_mfma.log
for rep in 1 2; do
for V in base noshared; do
  if [ $V = base ]; then L=""; else L=$R/gpurun_tmp_libs/libggq_$V.so; fi
  for Q in Q4_K Q5_K; do
  GGQ_HIP_LIB=$L timeout 300 python tools/mfma_linear_bench.py --qtype $Q --shapes 12288x3072,3072x12288,21504x3072 --m 1024,4608 --tiles 256 > $O/${V}_${Q}_$rep.json 2>> $O/err.log
  python - $O/${V}_${Q}_$rep.json $V $Q <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('%-9s %s' % (sys.argv[2], sys.argv[3]), [(r['weight'][:5], r['m'], r['fused tile=256'], r['dequant+F.linear']) for r in d['rows']])
PY
  done
done; done
