#!/bin/bash
# round 3, GPU session AB: with s_setprio in place, 8 waves (128 x 64 per wave, 2 per SIMD: a clean two-wave ping-pong) against the shipped 16 waves
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3ab; mkdir -p $O
for rep in 1 2; do
for WM in 4 2; do
  GGQ_HIP_LIB=$R/gpurun_tmp_libs/libggq_wmab.so GGQ_TILE_WM=$WM timeout 300 python tools/mfma_linear_bench.py --shapes 12288x3072,3072x12288,21504x3072 --m 1024,4608 --tiles 256 > $O/wm${WM}_$rep.json 2>> $O/err.log
  python - $O/wm${WM}_$rep.json $WM <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('WM', sys.argv[2], [(r['weight'], r['m'], r['fused tile=256'], r['dequant+F.linear']) for r in d['rows']])
PY
done; done
tail -2 $O/err.log
