#!/bin/bash
# round 3, GPU session V: layer-sized launch shapes re-swept for the production case -- bf16 result, write-through stores (Next #2 candidate (c))
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3v; mkdir -p $O
( timeout 900 tests/microbench/ggq_microbench ablayer3 > $O/ablayer3.txt 2> $O/ablayer3.err; echo "rc=$?" >> $O/ablayer3.err ); tail -2 $O/ablayer3.err
grep -c "^AB" $O/ablayer3.txt; cut -c1-150 $O/ablayer3.txt
