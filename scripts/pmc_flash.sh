# PMC passes (rocprofv3 --pmc, one counter group per run, no other tracing) over the shipped d=40 flash kernel
# (attn_flash_kernel<40,2,2,true,0>, q in the log2 domain) as launched by scripts/kbench.py --judged:
# 8 frames x 8 heads x 4096 queries x 8192 keys per launch.  Output: gpurun_out/pmc/flash_d40.json
mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/kbench.py --judged > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "attn_flash_kernelILi40ELi2ELi2ELb1ELi0E" in k and int(r["Grid_Size"]) == 8 * 8 * 16 * 256:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: sum(v) / len(v) for c, v in agg.items()}
res["_launches_sampled"] = {c: len(v) for c, v in agg.items()}
res["_note"] = ("per launch of attn_flash_kernel<40,W=2,QB=2,bias slot> on 8 frames x 8 heads x 4096 x 8192; FETCH_SIZE/WRITE_SIZE in KiB as "
                "rocprofv3 reports them (gfx950: FETCH_SIZE counts 64 B per 128-B request -> double it, MI355X_MICROARCH.md HBM section)")
json.dump(res, open("$OUT/flash_d40.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/*/p_kernel_trace.csv $OUT/*/*agent_info.csv
