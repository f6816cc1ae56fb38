#!/bin/bash
# round 3, GPU session AG: the shared-tile GEMM for every format at the production shape (12288 x 3072, 4608 rows, bf16)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3ag; mkdir -p $O
for Q in Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K IQ4_NL IQ4_XS; do
  timeout 200 python tools/mfma_linear_bench.py --qtype $Q --shapes 12288x3072 --m 4608 --tiles 256 > $O/$Q.json 2>> $O/err.log
  python - $O/$Q.json $Q <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['rows'][0]
print('%-7s tile %.1f us  %.0f TFLOP/s   unpack+hipBLASLt %.1f   dense-resident %.1f' % (sys.argv[2], r['fused tile=256'], r['GFLOP']/r['fused tile=256']*1e-3*1e3, r['dequant+F.linear'], r['F.linear dense-resident']))
PY
done
