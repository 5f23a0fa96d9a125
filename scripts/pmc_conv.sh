# PMC passes over conv3x3_kernel<2,2,32> at the 64x64 level (8 frames x 4096 px, 320 -> 320), scripts/kbench.py --conv64.
mkdir -p gpurun_out/pmc_conv; cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
for set in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/kbench.py --conv64 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv3x3_kernel" in r.get("Kernel_Name", ""):
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"grid_%s" % g: {c: sum(v) / len(v) for c, v in d.items()} for g, d in agg.items()}
res["_note"] = "per launch of conv3x3_kernel<2,2,32>, 320->320 at 64x64; grid 196608 = 8 frames (256 x 3 blocks of 256 threads), 393216 = 16 frames"
json.dump(res, open("$OUT/conv64.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/*/p_kernel_trace.csv $OUT/*/*agent_info.csv $OUT/*/p_counter_collection.csv
