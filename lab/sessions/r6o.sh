# round 6, session o: SD3.5's Q5_0 layers at 32 / 64 rows of x regressed in the token sweep (64 tokens 7.23 -> 8.18 ms): which change -- 8 waves below 256 workgroups, or the unaligned-row handling of 32-element formats?
O=gpurun_out/r6o; mkdir -p $O
L=$PWD/gpurun_tmp_libs
SH=7296x2432,2432x2432,9728x2432
for i in 1 2; do
  GGQ_HIP_LIB=$L/libggq_lab.so timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes $SH > $O/rule_$i.json 2>> $O/err.log
  GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF32_KW=4 timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes $SH > $O/kw4_$i.json 2>> $O/err.log
  GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF32_KW=8 timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes $SH > $O/kw8_$i.json 2>> $O/err.log
  GGQ_HIP_LIB=$L/libggq_oldalign.so GGQ_MF32_KW=4 timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes 7296x2560,2432x2560,9728x2560 > $O/oldalign_kw4_$i.json 2>> $O/err.log
  GGQ_HIP_LIB=$L/libggq_lab.so GGQ_MF32_KW=4 timeout 300 python tools/fused_sweep.py --qtype Q5_0 --m 32,64 --kernels mfma:0 --shapes 7296x2560,2432x2560,9728x2560 > $O/newalign_kw4_$i.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
tab={}
for f in sorted(glob.glob("gpurun_out/r6o/*_?.json")):
    v=f.split("/")[-1][:-5]
    for r in json.load(open(f))["rows"]: tab.setdefault((r["weight"],r["m"]),{})[v]=r.get("mfma:0")
for k,row in tab.items(): print(k,row)
PY
timeout 600 python tools/flux_forward_emulation.py --model sd35 --tokens 64 --fused-small-m --fused-mfma 256 --graph > $O/sd35_64.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/sd35_64.json')); print(d['ms_per_step_dequant_on_the_fly'], d['ms_per_step_dense_resident'], d['graph_replay_ms_per_step'])"
