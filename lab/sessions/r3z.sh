#!/bin/bash
# round 3, GPU session Z: s_setprio around the MFMA half (1) or the decode half (2) of the shared-tile GEMM's K-step, against the shipped kernel; two alternations
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O
( GGQ_HIP_LIB=$R/gpurun_tmp_libs/libggq_prio1.so timeout 300 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu -k "tile or 256" > $O/tests_prio1.log 2>&1; tail -2 $O/tests_prio1.log )
for rep in 1 2; do
for V in base prio1 prio2; do
  if [ $V = base ]; then L=""; else L=$R/gpurun_tmp_libs/libggq_$V.so; fi
  GGQ_HIP_LIB=$L timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288,21504x3072 --m 1024,4608 --tiles 256 > $O/${V}_$rep.json 2>> $O/err.log
  python - $O/${V}_$rep.json $V <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], [(r['weight'], r['m'], r['fused tile=256'], r['dequant+F.linear']) for r in d['rows']])
PY
done; done
