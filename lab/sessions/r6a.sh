# round 6, session a: VALU issue-rate probe (tools/probes/valu_rates.hip, built into gpurun_tmp_libs/)
O=gpurun_out/r6a; mkdir -p $O
timeout 300 gpurun_tmp_libs/valu_rates > $O/valu_rates.json 2> $O/valu_rates.err
tail -c 300 $O/valu_rates.json
