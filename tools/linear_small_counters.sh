# rocprofv3 --pmc passes (--kernel-trace only) over tools/linear_small_counters.py; prints mean counters per launch of ggq::linear_small by (M, rows)
R=$PWD; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/linpmc
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/linpmc/p$i -o p -- python $R/tools/linear_small_counters.py > $R/gpurun_out/linpmc/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/linpmc/p$i.log)"
done
cd $R; python - <<'PY'
import csv, glob, collections, re
tab=collections.OrderedDict(); dur=collections.defaultdict(list)
def key(name, grid):
    m=re.search(r"linear_small<ggq::\w+, \d, (\d)>",name)
    return "M"+m.group(1)+"/wg"+str(grid)
for d in sorted(glob.glob("gpurun_out/linpmc/p*/")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(d+"p_counter_collection.csv")):
        if "linear_small" in r["Kernel_Name"]:
            acc[(r["Counter_Name"],key(r["Kernel_Name"],int(r["Grid_Size"])//256))].append(float(r["Counter_Value"]))
    for (n,k),v in acc.items(): tab.setdefault(n,{})[k]=sum(v)/len(v)
    for r in csv.DictReader(open(d+"p_kernel_trace.csv")):
        if "linear_small" in r["Kernel_Name"]:
            g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])//256
            dur[key(r["Kernel_Name"],g)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cols=sorted(dur)
print("%-32s"%"counter"+"".join("%16s"%c for c in cols))
print("%-32s"%"duration_us"+"".join("%16.1f"%(sum(dur[c])/len(dur[c])) for c in cols))
for n,row in tab.items(): print("%-32s"%n+"".join("%16.4g"%row.get(c,float("nan")) for c in cols))
PY
