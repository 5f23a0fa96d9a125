mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep "^Counter_Name" | awk '{print $NF}' | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
export FZ_FLASH_WAVES=4
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/kbench.py --flash > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
fs=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc/$tag/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "attn_flash_kernelILi40" in k:
            agg[r["Counter_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
for c,d in agg.items():
    for g,v in d.items():
        print(c, "grid", g, "n", len(v), "mean", sum(v)/len(v))
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc/*/p_kernel_trace.csv
