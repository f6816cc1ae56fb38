#!/bin/bash
# round 3, GPU session AC: s_setprio around the MFMAs of the K-split kernel (<= 256 rows), two alternations
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3ac; mkdir -p $O
for rep in 1 2; do
for V in base mfprio; do
  if [ $V = base ]; then L=""; else L=$R/gpurun_tmp_libs/libggq_$V.so; fi
  GGQ_HIP_LIB=$L timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x3072,3072x12288,21504x3072 --m 32,64,128,256 --tiles 0 > $O/${V}_$rep.json 2>> $O/err.log
  python - $O/${V}_$rep.json $V <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], [(r['weight'][:5], r['m'], r['fused tile=auto']) for r in d['rows']])
PY
done; done
tail -2 $O/err.log
