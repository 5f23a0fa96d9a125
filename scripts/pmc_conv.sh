# PMC passes (rocprofv3 --pmc, one counter group per run, no other tracing) over the implicit-GEMM 3x3 convolution at the 64x64
# level (320 -> 320; 8 frames: 320x128 tiles, 16 frames: 320x256 tiles), as launched by scripts/kbench.py --conv64.
#   -> gpurun_out/pmc/conv64.json (per-launch averages per grid size)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc/conv64; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/g$i -o p -- python $R/scripts/kbench.py --conv64 > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/*counter_collection.csv") + glob.glob("$OUT/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "igemm_kernel" in k:
            agg[k[:60] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {g: {c: sum(v) / len(v) for c, v in sorted(d.items())} for g, d in agg.items()}
res["_note"] = "per launch, 3x3 conv 320->320 at 64x64: 8 frames = 320x128 tiles (256 workgroups), 16 frames = 320x256 tiles (256 workgroups)"
json.dump(res, open("$R/gpurun_out/pmc/conv64.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT
