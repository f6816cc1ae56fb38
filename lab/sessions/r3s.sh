#!/bin/bash
# round 3, GPU session S: smoke() + the whole -m gpu suite on the final tree
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3s; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log ); tail -3 $O/smoke.log
( timeout 1300 python -m pytest tests -x -q -m gpu --durations=8 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -16 $O/tests.log
