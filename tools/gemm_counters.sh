# rocprofv3 --pmc passes (--kernel-trace only) over tools/gemm_counters.py; prints mean counters per launch of ggq::linear_tile by workgroups.
# usage: bash tools/gemm_counters.sh <outdir under gpurun_out>
R=$PWD; O=$R/gpurun_out/${1:-gemmpmc}; cd /tmp && export TMPDIR=/tmp
mkdir -p $O
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/gemm_counters.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
cd $R; python - $O <<'PY'
import csv, glob, collections, sys
O=sys.argv[1]
tab=collections.OrderedDict(); dur=collections.defaultdict(list)
for d in sorted(glob.glob(O+"/p*/")):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "linear_tile" in r["Kernel_Name"]:
                acc[(r["Counter_Name"],"wg"+str(int(r["Grid_Size"])//512))].append(float(r["Counter_Value"]))
    for (n,k),v in acc.items(): tab.setdefault(n,{})[k]=sum(v)/len(v)
    for f in glob.glob(d+"**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "linear_tile" in r["Kernel_Name"]:
                g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])//512
                dur["wg"+str(g)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cols=sorted(dur)
print("%-32s"%"counter"+"".join("%16s"%c for c in cols))
print("%-32s"%"duration_us"+"".join("%16.1f"%(sum(dur[c])/len(dur[c])) for c in cols))
for n,row in tab.items(): print("%-32s"%n+"".join("%16.4g"%row.get(c,float("nan")) for c in cols))
PY
