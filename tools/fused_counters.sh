# rocprofv3 --pmc passes (--kernel-trace only) over tools/fused_counters.py "$1"; prints mean counters per launch by (kernel, grid x block).  Output dir: gpurun_out/fusedpmc$2
R=$PWD; cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/fusedpmc$2; mkdir -p $D
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAVES_EQ_64" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D/p$i -o p -- python $R/tools/fused_counters.py "$1" > $D/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $D/p$i.log)"
done
cd $R; python - "$D" <<'PY'
import csv, glob, collections, re, sys
D=sys.argv[1]
def key(name, gx, wg):
    m=re.search(r"ggq::(linear_\w+)<ggq::Fmt(\w+), ([^>]*)>", name)
    if not m: return None
    return f"{m.group(1)}<{m.group(3).replace(' ','')}>/g{gx}x{wg}"
tab=collections.OrderedDict(); dur=collections.defaultdict(list)
for d in sorted(glob.glob(D+"/p*/")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(d+"p_counter_collection.csv")):
        k=key(r["Kernel_Name"], int(r["Grid_Size"])//int(r["Workgroup_Size"]), r["Workgroup_Size"])
        if k: acc[(r["Counter_Name"],k)].append(float(r["Counter_Value"]))
    for (n,k),v in acc.items(): tab.setdefault(n,{})[k]=sum(v)/len(v)
    for r in csv.DictReader(open(d+"p_kernel_trace.csv")):
        wg=int(r["Workgroup_Size_X"])*int(r["Workgroup_Size_Y"])*int(r["Workgroup_Size_Z"])
        g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])//wg
        k=key(r["Kernel_Name"], g, wg)
        if k: dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cols=sorted(dur)
w=max(len(c) for c in cols)+2
print("%-32s"%"counter"+"".join(("%"+str(w)+"s")%c for c in cols))
print("%-32s"%"duration_us"+"".join(("%"+str(w)+".1f")%(sum(dur[c])/len(dur[c])) for c in cols))
for n,row in tab.items(): print("%-32s"%n+"".join(("%"+str(w)+".4g")%row.get(c,float("nan")) for c in cols))
PY
