# round 5, last session: the whole GPU suite + smoke() on the final tree
O=gpurun_out/r5z; mkdir -p $O
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8) > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
