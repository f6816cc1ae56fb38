#!/bin/bash
# round 3, GPU session AD: more cheap scheduling knobs of the shared-tile GEMM against the shipped build (setprio 1, pong mask 0xA, cross 0x100): setprio only around the MFMAs,
# 1 / 3 / (2 adjacent) of 4 waves per SIMD MFMA-first, nothing crossing the half-step line.  Two alternations.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3ai; mkdir -p $O
for rep in 1 2; do
for V in base xring0 cross180 prio3l pongC pong8; do
  if [ $V = base ]; then L=""; else L=$R/gpurun_tmp_libs/libggq_$V.so; fi
  GGQ_HIP_LIB=$L timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288,21504x3072 --m 1024,4608 --tiles 256 > $O/${V}_$rep.json 2>> $O/err.log
  python - $O/${V}_$rep.json $V <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('%-7s' % sys.argv[2], [(r['weight'][:5], r['m'], r['fused tile=256']) for r in d['rows']])
PY
done; done
