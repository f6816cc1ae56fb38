# round 6, session j: compile / autograd / doc tests on the GPU; emulated FLUX step at 64 / 128 tokens (graph) against the round-5 figures; mid-m sweep (T5 / SD3.5 shapes, 256 < m <= 1024)
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference.py tests/test_gpu_integration_doc.py tests/test_gpu_rows.py tests/test_gpu_linear.py -q -rs -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -8 $O/tests.log
for t in 64 128 256; do timeout 600 python tools/flux_forward_emulation.py --tokens $t --fused-small-m --fused-mfma 256 --graph > $O/flux_$t.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/flux_$t.json')); print($t, d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['graph_replay_ms_per_step'])"; done
timeout 900 python tools/fused_sweep.py --qtype Q4_K --m 256,384,512,768,1024 --kernels default,mfma:64,mfma:128,mfma:256 --shapes 4096x4096,10240x4096,4096x10240,2432x9728 > $O/midm_q4k.json 2> $O/midm_q4k.err
timeout 900 python tools/fused_sweep.py --qtype Q5_0 --m 256,384,512,768,1024 --kernels default,mfma:64,mfma:128 --shapes 7296x2432,2432x2432,9728x2432,14592x2432 > $O/midm_q50.json 2> $O/midm_q50.err
python - <<'PY'
import json
for f in ("midm_q4k","midm_q50"):
    for r in json.load(open(f"gpurun_out/r6j/{f}.json"))["rows"]:
        print(f, r["weight"], r["m"], {k:v for k,v in r.items() if k in ("default","mfma:64","mfma:128","mfma:256")})
PY
