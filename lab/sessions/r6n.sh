# round 6, session n: K split across workgroups (ggq_linear_mfma_ws) on the weights with few, long rows -- slice count sweep (lab build, GGQ_MF32_ZS) and the shipped rule; emulated FLUX step at 64 / 128 tokens
O=gpurun_out/r6n; mkdir -p $O
timeout 600 python tools/fused_sweep.py --m 16,32,64,128,256 --kernels mfma:0 --shapes 3072x12288,3072x15360,3072x3072,4096x10240,9216x3072 > $O/shipped.json 2> $O/shipped.err; cat $O/shipped.err | cut -c1-160
export GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_lab.so
for zs in 1 2 4 6 8 12 16; do
  GGQ_MF32_ZS=$zs timeout 600 python tools/fused_sweep.py --m 32,64,128,256 --kernels mfma:0 --shapes 3072x12288,3072x15360,3072x3072,9216x3072 > $O/zs$zs.json 2> $O/zs$zs.err
done
unset GGQ_HIP_LIB
python - <<'PY'
import json
tab={}
for zs in (1,2,4,6,8,12,16):
    for r in json.load(open(f"gpurun_out/r6n/zs{zs}.json"))["rows"]: tab.setdefault((r["weight"],r["m"]),{})[zs]=r.get("mfma:0")
for k,row in tab.items(): print(k,row)
PY
for t in 64 128; do timeout 600 python tools/flux_forward_emulation.py --tokens $t --fused-small-m --fused-mfma 256 --graph > $O/flux_$t.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/flux_$t.json')); print($t, d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['graph_replay_ms_per_step'])"; done
