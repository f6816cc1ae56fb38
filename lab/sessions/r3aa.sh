#!/bin/bash
# round 3, GPU session AA: the shared-tile GEMM with s_setprio (new default): tests + the GEMM table
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3aa; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_linear.py -x -q -m gpu > $O/tests_mfma.log 2>&1; echo "rc=$?" >> $O/tests_mfma.log ); tail -3 $O/tests_mfma.log
( timeout 400 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x3072,3072x12288,21504x3072 --m 64,256,1024,4608 --tiles 0,128,256 > $O/gemm_bench.json 2> $O/gemm_bench.err; echo "rc=$?" >> $O/gemm_bench.err ); tail -1 $O/gemm_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3aa/gemm_bench.json'))
for r in d['rows']: print(r['weight'], r['m'], r.get('fused tile=256'), r['dequant+F.linear'], r['F.linear dense-resident'], r['fused_best_TFLOPs'])
PY
