# PMC passes (rocprofv3 --pmc, one counter group per run, no other tracing) over ONE variant of the d=40 flash kernel as
# launched by scripts/flash_ab.hip (8 frames x 8 heads x 4096 queries x 8192 keys per launch, q in the log2 domain).
#   bash scripts/pmc_flash.sh <variant index of flash_ab> <tag>        -> gpurun_out/pmc/<tag>.json (per-launch averages)
V=${1:-1}; TAG=${2:-flash_d40}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc/$TAG; mkdir -p $OUT $R/build_tmp
[ -x $R/build_tmp/flash_ab ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFLASH_AB_OLD -o $R/build_tmp/flash_ab $R/scripts/flash_ab.hip
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_SALU" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
  "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/g$i -o p -- $R/build_tmp/flash_ab $V > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
names = set()
for f in glob.glob("$OUT/g*/*counter_collection.csv") + glob.glob("$OUT/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "attn_flash" in k:
            names.add(k[:80])
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: sum(v) / len(v) for c, v in sorted(agg.items())}
res["_launches_sampled"] = {c: len(v) for c, v in agg.items()}
res["_kernels"] = sorted(names)
res["_note"] = ("per launch of flash_ab variant $V on 8 frames x 8 heads x 4096 x 8192; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them "
                "(gfx950: FETCH_SIZE counts 64 B per 128-B request -> double it, MI355X_MICROARCH.md HBM section)")
json.dump(res, open("$R/gpurun_out/pmc/$TAG.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT
